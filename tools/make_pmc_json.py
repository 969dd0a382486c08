#!/usr/bin/env python3
"""HBM traffic per kernel launch of the dense C2 step from two rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE cannot share a pass:
MI355X_MICROARCH.md, PMC slots), with the gfx950 correction of that guide (FETCH_SIZE reports half of a wide coalesced read: doubled).

  rocprofv3 --pmc FETCH_SIZE --kernel-trace -d gpurun_out/pmc_f -- python3 tools/prof_dense.py 4096 4096 0 3 0
  rocprofv3 --pmc WRITE_SIZE --kernel-trace -d gpurun_out/pmc_w -- python3 tools/prof_dense.py 4096 4096 0 3 0
  python tools/make_pmc_json.py gpurun_out/pmc_f gpurun_out/pmc_w 4096 4096 0 "<collected: date / commit>" > profiles/r03_pmc_dense_c2.json
"""
import glob
import json
import sqlite3
import sys


def per_kernel(path, counter):
    out = {}
    for db in sorted(glob.glob(path + "/**/*_results.db", recursive=True)):
        con = sqlite3.connect(db)
        cur = con.cursor()
        tables = [r[0] for r in cur.execute("select name from sqlite_master where type in ('table','view')")]
        if "counters_collection" not in tables:
            continue
        cols = [r[1] for r in cur.execute("pragma table_info(counters_collection)")]
        kcol = "kernel_name" if "kernel_name" in cols else "name"
        for k, v, n in cur.execute(f"select {kcol}, sum(value), count(*) from counters_collection where counter_name = ? group by {kcol}", (counter,)):
            e = out.setdefault(k, [0.0, 0])
            e[0] += v; e[1] += n
    return out


def main():
    f_dir, w_dir, n, m, p = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
    collected = sys.argv[6] if len(sys.argv) > 6 else ""
    F, W = per_kernel(f_dir, "FETCH_SIZE"), per_kernel(w_dir, "WRITE_SIZE")
    kernels = {}
    for k in sorted(set(F) | set(W)):
        if "pq::" not in k:
            continue
        f, fn = F.get(k, [0.0, 0]); w, wn = W.get(k, [0.0, 0])
        kernels[k] = {"dispatches": max(fn, wn), "fetch_bytes_per_launch": 2.0 * f * 1024.0 / max(fn, 1), "write_bytes_per_launch": w * 1024.0 / max(wn, 1)}

    def tot(sub):
        return sum(v["fetch_bytes_per_launch"] + v["write_bytes_per_launch"] for k, v in kernels.items() if sub in k)

    def per_step(subs, steps_of):
        # bytes of all launches of the kernels matching `subs` divided by the number of steps (= dispatches of `steps_of`'s first kernel / its launches per step)
        return sum((v["fetch_bytes_per_launch"] + v["write_bytes_per_launch"]) * v["dispatches"] for k, v in kernels.items() if any(s in k for s in subs)) / steps_of

    asm_main = [v for k, v in kernels.items() if "k_syrk_lower<0, 2, 2>" in k]
    steps = asm_main[0]["dispatches"] / 2.0 if asm_main else 1.0  # main + split-K tail launch per assembly
    out = {"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) -- python3 tools/prof_dense.py", "collected": collected,
           "correction": "gfx950: FETCH_SIZE counts 64 B per 128-B request on wide coalesced reads -> doubled (MI355X_MICROARCH.md, HBM section); WRITE_SIZE as reported",
           "n": n, "m": m, "p": p, "steps_profiled": steps, "kernels": kernels,
           "per_launch": {"assembly": {"traffic_bytes": per_step(["k_syrk_lower<0, 2, 2>", "k_syrk_tail_reduce<0>"], steps),
                                       "note": "main launch + split-K tail + tail reduce = one assembly"},
                          # round 3: one persistent launch per factorisation (k_chol_persistent); before: the average over the 31 fused launches
                          "panel_update": ({"traffic_bytes": tot("k_chol_persistent"), "note": "k_chol_persistent: all rounds of one factorisation (a single launch)"}
                                           if tot("k_chol_persistent") > 0 else
                                           {"traffic_bytes": tot("k_syrk_lower<3, 4, 2>"), "note": "average over the 31 fused trailing-update launches of a factorisation"}),
                          "panel_solve": {"traffic_bytes": tot("k_trsm_panel")},
                          "triangular_sweep": {"traffic_bytes": 0.5 * tot("k_trsv_persistent")}}}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
