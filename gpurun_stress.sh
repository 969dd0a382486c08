cd $GRAFT_REPO_ROOT
ulimit -v 60000000
timeout 600 python3 tools/stress_shapes.py 2>&1 | tail -16 | cut -c1-250
