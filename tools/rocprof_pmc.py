#!/usr/bin/env python3
"""Sums rocprofv3 --pmc counters per kernel from rocpd sqlite dbs:  python tools/rocprof_pmc.py <dir> [kernel-substring]"""
import glob
import sqlite3
import sys

path = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
for db in sorted(glob.glob(path + "/**/*_results.db", recursive=True)):
    con = sqlite3.connect(db)
    cur = con.cursor()
    tables = [r[0] for r in cur.execute("select name from sqlite_master where type in ('table','view')")]
    view = "counters_collection" if "counters_collection" in tables else None
    if view is None:
        print("no counters in", db)
        continue
    cols = [r[1] for r in cur.execute(f"pragma table_info({view})")]
    kcol = "kernel_name" if "kernel_name" in cols else "name"
    rows = list(cur.execute(f"select {kcol}, counter_name, sum(value), count(*) from {view} group by {kcol}, counter_name"))
    print("#", db)
    for k, c, v, n in rows:
        if flt in k:
            print(f"{k[:60]:60s} {c:28s} {v:18.0f}  (rows {n})")
