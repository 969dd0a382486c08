#!/usr/bin/env python3
"""Per-launch timeline of the LAST factorisation + solve recorded in a rocprofv3 rocpd database (tools/prof_sparse.py under
`rocprofv3 --kernel-trace`): kernel, grid, duration, and per-kernel totals.
usage: python tools/rocpd_timeline.py results.db [--all]"""
import collections
import re
import sqlite3
import sys

c = sqlite3.connect(sys.argv[1])
tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if "kernel_dispatch" in t][0]
ks = [t for t in tabs if "kernel_symbol" in t][0]
rows = list(c.execute(f"select s.kernel_name,d.start,d.end,d.grid_size_x,d.workgroup_size_x,d.grid_size_y from {kd} d join {ks} s on d.kernel_id=s.id order by d.start"))


def short(n):
    m = re.search(r"k_[a-z_0-9]+", n)
    return m.group(0) if m else n[:30]


idx = [i for i, r in enumerate(rows) if "subtree_factor" in r[0]]
a = idx[-1]
seg = rows[a:]
t0 = seg[0][1]
agg = collections.defaultdict(lambda: [0, 0.0])
for r in seg:
    agg[short(r[0])][0] += 1
    agg[short(r[0])][1] += (r[2] - r[1]) / 1e3
    if "--all" in sys.argv:
        print(f"{(r[1]-t0)/1e3:9.1f} {short(r[0]):28s} grid={r[3]//r[4]:5d} x{r[5]:3d} x{r[4]:4d} dur={(r[2]-r[1])/1e3:7.1f}")
print("span us", (seg[-1][2] - t0) / 1e3)
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f"{k:32s} n={v[0]:5d} sum={v[1]:9.1f} us")
