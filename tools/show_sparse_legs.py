"""Prints the sparse legs (value, ms per step, factor / backend-solve ms) of a bench.py JSON line read from stdin."""
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
for k,v in d["sparse_kkt"].items():
    print(k, {q:round(v[q],4) for q in ("value","ms_per_step","factor_ms","backend_solve_ms") if q in v})
