cd $GRAFT_REPO_ROOT
ulimit -v 100000000
mkdir -p gpurun_out
timeout 1500 python3 tools/exp_cond_parity.py 4 > gpurun_out/r04_cond_parity.txt 2>/dev/null
tail -3 gpurun_out/r04_cond_parity.txt | cut -c1-600
grep -c "differs" gpurun_out/r04_cond_parity.txt
