cd $GRAFT_REPO_ROOT
ulimit -v 100000000
timeout 1500 python3 -m pytest tests/test_partition.py -x -q 2>&1 | tail -8
