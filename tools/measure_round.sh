# How the records under profiles/r05_* were produced (run on the GPU box through gpurun): GPU test suite, bench line (the last stdout line must stay under 6 KB),
# rocprofv3 kernel stats of the bench command, whole-solve parity sweep (device vs oracle, per-iteration tables compared bitwise), timings of the reference-order engine.
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
cd /tmp; export TMPDIR=/tmp
O=$R/gpurun_out
cd $R
ulimit -v 150000000
timeout 2400 python3 -m pytest tests -m gpu -q 2>&1 | tail -8 > $O/r05_gpu_tests.txt
tail -3 $O/r05_gpu_tests.txt
timeout 1500 python3 bench.py > $O/r05_bench_stdout.txt 2> $O/r05_bench_n1.err
echo "bench rc=$?"
tail -1 $O/r05_bench_stdout.txt > $O/r05_bench_n1.json
echo "last line bytes: $(tail -1 $O/r05_bench_stdout.txt | wc -c) (limit 6144)"
cp bench_details.json $O/r05_bench_details.json 2>/dev/null
rm -rf $O/ks; timeout 900 rocprofv3 --kernel-trace --stats -d $O/ks -o bench -- python3 bench.py --no-size-sweep > $O/r05_bench_under_rocprof.txt 2> $O/r05_bench_under_rocprof.err
echo "rocprof bench rc=$?"
python3 tools/rocprof_summary.py $O/ks 40 > $O/r05_bench_kernel_stats.txt 2>&1
rm -rf $O/ks
timeout 900 python3 tools/trace_diff.py --all 2>/dev/null > $O/r05_whole_solve_parity.txt; tail -1 $O/r05_whole_solve_parity.txt
for ks in 2 3 4; do timeout 900 python3 tools/trace_diff.py --all --ks=$ks 2>/dev/null > $O/r05_whole_solve_parity_ks$ks.txt; tail -1 $O/r05_whole_solve_parity_ks$ks.txt; done
timeout 300 python3 tools/time_exact.py 2>/dev/null > $O/r05_exact_engine_timing.txt; tail -16 $O/r05_exact_engine_timing.txt | cut -c1-220
timeout 300 python3 tools/sqp_benchmarks.py > $O/r05_sqp_benchmarks.txt 2>/dev/null
tail -2 $O/r05_sqp_benchmarks.txt | cut -c1-200
timeout 600 python3 tools/mm_timing.py mm_ nl_ > $O/r05_mm_timing.txt 2>/dev/null; tail -17 $O/r05_mm_timing.txt | cut -c1-160
