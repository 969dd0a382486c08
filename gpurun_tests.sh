cd $GRAFT_REPO_ROOT
ulimit -v 80000000
timeout 900 python3 -m pytest tests/test_sparse_gpu.py tests/test_sparse_variants_gpu.py tests/test_abi.py -x -q 2>&1 | tail -5
