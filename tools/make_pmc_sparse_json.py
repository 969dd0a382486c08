#!/usr/bin/env python3
"""HBM traffic per factorisation / per backend solve of a sparse workload from two rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE cannot share a pass),
gfx950 correction of MI355X_MICROARCH.md (FETCH_SIZE reports 64 B per 128-B request on wide coalesced reads: doubled, the raw figure beside it).

  rocprofv3 --pmc FETCH_SIZE --kernel-trace -d gpurun_out/pmc_sf -- python3 tools/prof_sparse.py --fixture mm_CONT-201 --no-oracle --reps 5
  rocprofv3 --pmc WRITE_SIZE --kernel-trace -d gpurun_out/pmc_sw -- python3 tools/prof_sparse.py --fixture mm_CONT-201 --no-oracle --reps 5
  python tools/make_pmc_sparse_json.py gpurun_out/pmc_sf gpurun_out/pmc_sw <key> "<workload>" <nnzL> <N> <nnzK> > profiles/r04_pmc_sparse_<key>.json
"""
import glob, json, re, sqlite3, sys

FACTOR = ["k_subtree_factor_lds", "k_subtree_factor_pk", "k_top_factor", "k_top_factor_walk", "k_front_factor", "k_big_zero", "k_big_assemble", "k_big_extend_add", "k_potrf_trsm_fronts",
          "k_potrf_diag_fronts", "k_trsm_panel_fronts", "k_front_panel_step", "k_syrk_half_fronts", "k_syrk_lower_fronts"]
SOLVE = ["k_subtree_fwd_wave", "k_subtree_bwd_wave", "k_front_fwd_wide", "k_front_bwd_wide", "k_front_fwd_rows", "k_front_bwd_cols", "k_level_fwd_mixed", "k_level_bwd_mixed", "k_scale",
         "k_perm_gather", "k_perm_scatter"]


def short(name):
    m = re.search(r"(k_[a-z0-9_]+)", name)
    return m.group(1) if m else name


def per_kernel(path, counter):
    out = {}
    for db in sorted(glob.glob(path + "/**/*_results.db", recursive=True)):
        cur = sqlite3.connect(db).cursor()
        tables = [r[0] for r in cur.execute("select name from sqlite_master where type in ('table','view')")]
        if "counters_collection" not in tables:
            continue
        cols = [r[1] for r in cur.execute("pragma table_info(counters_collection)")]
        kcol = "kernel_name" if "kernel_name" in cols else "name"
        for k, v, n in cur.execute(f"select {kcol}, sum(value), count(*) from counters_collection where counter_name = ? group by {kcol}", (counter,)):
            e = out.setdefault(short(k), [0.0, 0])
            e[0] += v; e[1] += n
    return out


def main():
    f_dir, w_dir, key, workload, nnzL, N, nnzK = sys.argv[1], sys.argv[2], sys.argv[3], sys.argv[4], float(sys.argv[5]), float(sys.argv[6]), float(sys.argv[7])
    F, W = per_kernel(f_dir, "FETCH_SIZE"), per_kernel(w_dir, "WRITE_SIZE")
    pk = {k: {"FETCH_SIZE_sum_KB": F.get(k, [0, 0])[0], "WRITE_SIZE_sum_KB": W.get(k, [0, 0])[0], "dispatches": max(F.get(k, [0, 0])[1], W.get(k, [0, 0])[1])} for k in sorted(set(F) | set(W)) if k.startswith("k_")}
    nfac = pk.get("k_big_zero", pk.get("k_subtree_factor_pk", pk.get("k_subtree_factor_lds", {"dispatches": 1})))["dispatches"]
    nsol = pk.get("k_perm_scatter", {"dispatches": 1})["dispatches"]

    def stage(names, count, alg):
        fr = sum(pk[k]["FETCH_SIZE_sum_KB"] for k in names if k in pk) * 1024.0 / max(count, 1)
        wr = sum(pk[k]["WRITE_SIZE_sum_KB"] for k in names if k in pk) * 1024.0 / max(count, 1)
        return {"fetch_bytes_raw": fr, "fetch_bytes_x2": 2 * fr, "write_bytes": wr, "traffic_bytes": 2 * fr + wr, "algorithmic_bytes": alg, "launches_profiled": count}
    out = {"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, --kernel-trace) -- python3 tools/prof_sparse.py; per kernel by tools/make_pmc_sparse_json.py",
           "units": "per_kernel_KB: counter sums over the run in KB as rocprofv3 reports them; *_per_launch: bytes per factorisation / per backend solve",
           "correction": "gfx950: FETCH_SIZE counts 64 B per 128-B request on wide coalesced reads (MI355X_MICROARCH.md, HBM section) -> doubled in traffic_bytes, raw figure beside it; WRITE_SIZE as reported",
           key: {"workload": workload, "per_kernel_KB": pk, "factor_kernels": [k for k in FACTOR if k in pk], "solve_kernels": [k for k in SOLVE if k in pk],
                 "factor_per_launch": stage(FACTOR, nfac, 12.0 * nnzK + 12.0 * nnzL + 24.0 * N), "solve_per_launch": stage(SOLVE, nsol, 24.0 * nnzL + 48.0 * N)}}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
