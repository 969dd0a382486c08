mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
cd $R
ulimit -v 150000000
timeout 2400 python3 -m pytest tests -m gpu -x -q 2>&1 | tail -8 > $O/r04_tests_final.txt
tail -3 $O/r04_tests_final.txt
timeout 300 python3 -c "
import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
timeout 1500 python3 bench.py > $O/r04_bench_n1.json 2> $O/r04_bench_n1.err
echo "bench rc=$?"
