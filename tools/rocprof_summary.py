#!/usr/bin/env python3
"""Prints the per-kernel summary (calls, total us, avg us, %) from a rocprofv3 rocpd sqlite db."""
import glob
import sqlite3
import sys

path = sys.argv[1]
dbs = glob.glob(path + "/**/*_results.db", recursive=True) if not path.endswith(".db") else [path]
for db in dbs:
    cur = sqlite3.connect(db).cursor()
    rows = list(cur.execute("select name, total_calls, total_duration, average, percentage from top_kernels"))
    print(f"# {db}")
    print(f"{'kernel':70s} {'calls':>7s} {'total_us':>12s} {'avg_us':>10s} {'pct':>6s}")
    for name, calls, tot, avg, pct in rows[: int(sys.argv[2]) if len(sys.argv) > 2 else 30]:
        nm = name if len(name) <= 70 else name[:67] + "..."
        print(f"{nm:70s} {calls:7d} {tot:12.1f} {avg:10.2f} {pct:6.2f}")
