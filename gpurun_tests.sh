cd $GRAFT_REPO_ROOT
ulimit -v 80000000
timeout 900 python3 -m pytest tests/test_dense_gpu.py -x -q -s -k "recorded_ipm_states" 2>&1 | grep -v "^$" | tail -8 | cut -c1-250
