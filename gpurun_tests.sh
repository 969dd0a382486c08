cd $GRAFT_REPO_ROOT
ulimit -v 100000000
timeout 900 python3 -m pytest tests/test_sparse_variants_gpu.py -x -q 2>&1 | tail -3
timeout 600 python3 tools/prof_sparse.py --spread 300 --row-nnz 10 --no-oracle --reps 5 2>&1 | grep "^device: factor" | cut -c1-200
timeout 600 python3 tools/prof_sparse.py --fixture mm_CONT-201 --no-oracle --reps 10 2>&1 | grep "^device: factor" | cut -c1-200
