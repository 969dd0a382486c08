# How the records under profiles/r04_* were produced (run on the GPU box through gpurun): GPU test suite, bench line, rocprofv3 kernel stats of the bench
# command, PMC passes of the two sparse workloads (separate FETCH_SIZE / WRITE_SIZE passes), the SQP benchmark shapes.
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
cd /tmp; export TMPDIR=/tmp
O=$R/gpurun_out
cd $R
ulimit -v 150000000
timeout 2400 python3 -m pytest tests -m gpu -x -q 2>&1 | tail -8 > $O/r04_tests_final.txt
tail -4 $O/r04_tests_final.txt
timeout 1500 python3 bench.py > $O/r04_bench_n1.json 2> $O/r04_bench_n1.err
echo "bench rc=$?"
rm -rf $O/ks; timeout 900 rocprofv3 --kernel-trace --stats -d $O/ks -o bench -- python3 bench.py --no-size-sweep > $O/r04_bench_under_rocprof.json 2> $O/r04_bench_under_rocprof.err
echo "rocprof bench rc=$?"
python3 tools/rocprof_summary.py $O/ks 40 > $O/r04_bench_kernel_stats.txt 2>&1
rm -rf $O/ks
for w in cont201 c3_wide; do
  if [ $w = cont201 ]; then A="--fixture mm_CONT-201"; else A="--spread 300 --row-nnz 10"; fi
  rm -rf $O/pmc_sf $O/pmc_sw
  timeout 400 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $O/pmc_sf -- python3 tools/prof_sparse.py $A --no-oracle --reps 5 > $O/pmc_$w.txt 2>&1; echo "pmc f $w rc=$?"
  timeout 400 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $O/pmc_sw -- python3 tools/prof_sparse.py $A --no-oracle --reps 5 > /dev/null 2>&1; echo "pmc w $w rc=$?"
  if [ $w = cont201 ]; then
    python3 tools/make_pmc_sparse_json.py $O/pmc_sf $O/pmc_sw sparse_cont201 "Maros-Meszaros CONT-201, sparse_ldlt (round 4, final code)" 3578517 80595 279794 > $O/r04_pmc_sparse_cont201.json
  else
    python3 tools/make_pmc_sparse_json.py $O/pmc_sf $O/pmc_sw sparse_c3_wide "C3 recipe, rows of 10 nonzeros in 300-variable windows, sparse_ldlt on its nested-dissection tree (round 4, final code)" 45749769 100000 742599 > $O/r04_pmc_sparse_c3_wide.json
  fi
done
rm -rf $O/pmc_sf $O/pmc_sw
timeout 300 python3 tools/sqp_benchmarks.py > $O/r04_sqp_benchmarks.txt 2>/dev/null
tail -2 $O/r04_sqp_benchmarks.txt | cut -c1-200
