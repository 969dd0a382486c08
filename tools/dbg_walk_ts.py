#!/usr/bin/env python3
"""Reads the timestamp dump of the flag-ordered forward sweep (PIQP_AMD_DEBUG=dbg_ts=<file>) and prints when workgroups start / stop waiting / end."""
import sys
import numpy as np
f = open(sys.argv[1], "rb").read()
nw = int(np.frombuffer(f[:4], dtype=np.int32)[0])
ts = np.frombuffer(f[4:4 + 24 * nw], dtype=np.int64).reshape(nw, 3).astype(np.float64)
b = 4 + 24 * nw
lo = np.frombuffer(f[b:b + 4 * nw], dtype=np.int32)
hi = np.frombuffer(f[b + 4 * nw:b + 8 * nw], dtype=np.int32)
t0 = ts[:, 0].min()
us = (ts - t0) / 100.0  # 100 MHz
print("walks", nw, " kernel span %.1f us" % us[:, 2].max())
q = np.linspace(0, nw - 1, 21).astype(int)
print(" idx    len   start  waitdone   end   (us)")
for i in q:
    print("%6d %4d %8.1f %8.1f %8.1f" % (i, hi[i] - lo[i] + 1, us[i, 0], us[i, 1], us[i, 2]))
dur = us[:, 2] - us[:, 0]
wait = us[:, 1] - us[:, 0]
print("mean duration %.2f us, mean first wait %.2f us, mean work after first wait %.2f us" % (dur.mean(), wait.mean(), (us[:, 2] - us[:, 1]).mean()))
print("start time percentiles (us):", np.percentile(us[:, 0], [50, 90, 99, 100]).round(1))
order = np.argsort(us[:, 2])
print("last 5 to end:", [(int(i), round(us[i, 0], 1), round(us[i, 1], 1), round(us[i, 2], 1), int(hi[i] - lo[i] + 1)) for i in order[-5:]])

