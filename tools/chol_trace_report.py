#!/usr/bin/env python3
"""Reads the raw timeline of one k_chol_persistent launch (PIQP_AMD_DEBUG=chol_trace + PIQP_AMD_CHOL_TRACE_FILE=<file>) and prints where the workgroups' time went:
per task kind the time spent working (inputs -> done) and parked inside a task (drawn -> inputs), the number of busy workgroups over time, and the chain's pace.
   python tools/chol_trace_report.py <file> [grid]"""
import sys

import numpy as np

KIND = {0: "crew helper", 1: "crew owner", 2: "panel row", 3: "bulk tile", 4: "diag half", 5: "diag half", 7: "far visit", 8: "assembly token"}


def main():
    rows = np.loadtxt(sys.argv[1])
    grid = int(sys.argv[2]) if len(sys.argv) > 2 else 256
    kind, rnd = rows[:, 1].astype(int), rows[:, 2].astype(int)
    drawn, inp, done = rows[:, 6], rows[:, 7], rows[:, 8]
    end = done.max()
    print(f"{len(rows)} tickets, launch {end:.0f} us, {grid} workgroups = {grid * end / 1e3:.0f} CU.ms available")
    tot_work = tot_wait = 0.0
    for k in sorted(set(kind)):
        sel = kind == k
        work, wait = (done[sel] - inp[sel]).clip(0).sum(), (inp[sel] - drawn[sel]).clip(0).sum()
        tot_work += work; tot_wait += wait
        print(f"  {KIND.get(k, k):15s} {sel.sum():5d} tasks: working {work / 1e3:7.1f} CU.ms (avg {work / max(sel.sum(), 1):6.1f} us), parked inside {wait / 1e3:7.1f} CU.ms (avg {wait / max(sel.sum(), 1):6.1f} us)")
    idle = grid * end - tot_work - tot_wait
    print(f"  working {tot_work / 1e3:.0f} CU.ms = {100 * tot_work / (grid * end):.0f} %, parked inside tasks {tot_wait / 1e3:.0f} CU.ms = {100 * tot_wait / (grid * end):.0f} %, "
          f"between tasks (ticket draw, gate, exit) {idle / 1e3:.0f} CU.ms = {100 * idle / (grid * end):.0f} %")
    # busy workgroups over time
    edges = np.linspace(0, end, 21)
    print("  time slice [us]      working  parked  (average workgroups)")
    for a, b in zip(edges[:-1], edges[1:]):
        w = (np.minimum(done, b) - np.maximum(inp, a)).clip(0).sum() / (b - a)
        p = (np.minimum(inp, b) - np.maximum(drawn, a)).clip(0).sum() / (b - a)
        print(f"  {a:7.0f} - {b:7.0f}     {w:6.1f}  {p:6.1f}")
    own = kind == 1
    o = sorted(zip(rnd[own], done[own]))
    print("  crew owner done at: " + " ".join(f"{int(r)}:{d:.0f}" for r, d in o))


if __name__ == "__main__":
    main()
