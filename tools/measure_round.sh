# How the records under profiles/r06_* were produced (run on the GPU box through gpurun): GPU test suite, bench line (the last stdout line must stay under 6 KB),
# rocprofv3 kernel stats of the bench command, whole-solve parity sweep (device vs oracle, per-iteration tables compared bitwise), timings of the reference-order engine.
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
cd /tmp; export TMPDIR=/tmp
O=$R/gpurun_out
cd $R
ulimit -v 150000000
timeout 2400 python3 -m pytest tests -m gpu -q -n 2 2>&1 | tail -8 > $O/r06_gpu_tests.txt
tail -3 $O/r06_gpu_tests.txt
timeout 1500 python3 bench.py > $O/r06_bench_stdout.txt 2> $O/r06_bench_n1.err
echo "bench rc=$?"
tail -1 $O/r06_bench_stdout.txt > $O/r06_bench_n1.json
echo "last line bytes: $(tail -1 $O/r06_bench_stdout.txt | wc -c) (limit 6144)"
cp bench_details.json $O/r06_bench_details.json 2>/dev/null
rm -rf $O/ks; timeout 900 rocprofv3 --kernel-trace --stats -d $O/ks -o bench -- python3 bench.py --no-size-sweep > $O/r06_bench_under_rocprof.txt 2> $O/r06_bench_under_rocprof.err
echo "rocprof bench rc=$?"
python3 tools/rocprof_summary.py $O/ks 40 > $O/r06_bench_kernel_stats.txt 2>&1
rm -rf $O/ks
timeout 900 python3 tools/trace_diff.py --all 2>/dev/null > $O/r06_whole_solve_parity.txt; tail -1 $O/r06_whole_solve_parity.txt
for ks in 2 3 4; do timeout 900 python3 tools/trace_diff.py --all --ks=$ks 2>/dev/null > $O/r06_whole_solve_parity_ks$ks.txt; tail -1 $O/r06_whole_solve_parity_ks$ks.txt; done
timeout 300 python3 tools/time_exact.py 2>/dev/null > $O/r06_exact_engine_timing.txt; tail -16 $O/r06_exact_engine_timing.txt | cut -c1-220
timeout 300 python3 tools/sqp_benchmarks.py > $O/r06_sqp_benchmarks.txt 2>/dev/null
tail -2 $O/r06_sqp_benchmarks.txt | cut -c1-200
timeout 600 python3 tools/mm_timing.py mm_ nl_ > $O/r06_mm_timing.txt 2>/dev/null; tail -17 $O/r06_mm_timing.txt | cut -c1-160
timeout 900 python3 tools/dense_sparse_solver_benchmark.py > $O/r06_dense_sparse_solver_benchmark.txt 2>/dev/null; tail -4 $O/r06_dense_sparse_solver_benchmark.txt | cut -c1-160
# stage partition, ranks sharing this one GPU through gloo (no scaling can be read off these: they show what each rank evaluates and that the results are bitwise)
for b in multistage ldlt_cond; do
  timeout 600 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node=2 --master-addr 127.0.0.1 --master-port 29811 tools/dist_c5.py --stages 800 --steps 3 --warmup 1 --backend $b --full-solve --refine 2>/dev/null | grep '^{' | tail -1 > $O/r06_sharded_solve_${b}_world2.json
  python3 -c "
import json,sys
d=json.load(open('$O/r06_sharded_solve_${b}_world2.json'))
print('$b world 2: bitwise', d['bitwise_equal_all_ranks'], 'residual eval ms single / partitioned', round(d['residual_eval_ms_single_gpu'],3), round(d['residual_eval_ms_partitioned'],3), 'step ms single / partitioned', round(d['single_gpu_ms_per_step'],3), round(d['ms_per_step'],3), d['sharded_solve'])
" | cut -c1-400
done
# the dense sweeps: block inverses against the substitution form (residuals, agreement), the forward sweep's timeline on the 100 MHz clock, time per solve
timeout 600 python3 tools/chk_dd_sweeps.py 1024 1100 2048 4096 2>&1 | tail -9 | cut -c1-230 > $O/r06_dense_sweeps.txt
for v in inv_sweeps=1 inv_sweeps=0; do
  PIQP_AMD_DEBUG=$v timeout 300 python3 tools/time_sweeps.py 2>&1 | tail -1 >> $O/r06_dense_sweeps.txt
  PIQP_AMD_DEBUG=$v,trsv_ts CHK_TAG=t timeout 300 python3 tools/chk_dd_sweeps.py --child 4096 2>&1 | grep -m1 -A3 "forward sweep" | cut -c1-700 >> $O/r06_dense_sweeps.txt
done
tail -12 $O/r06_dense_sweeps.txt | cut -c1-200
# PMC passes (FETCH_SIZE / WRITE_SIZE apart) of the dense C2 step, the C4 batch and CONT-201: see the commands in profiles/r06_pmc_*.json ("source")
# round 6 additions: the dense Maros-Meszaros sweep, the fixtures above 8192 rows through the reference-order engine, the stall soak
timeout 900 python3 tools/dense_mm_parity.py > $O/r06_dense_mm_parity.txt 2>/dev/null; tail -1 $O/r06_dense_mm_parity.txt
timeout 900 python3 tools/time_big_engines.py > $O/r06_big_engines.txt 2>/dev/null; tail -3 $O/r06_big_engines.txt
timeout 900 python3 tools/soak_stalls.py 2000 > $O/r06_soak_dense.txt 2>/dev/null; grep -c "above 3 x median: 0" $O/r06_soak_dense.txt
timeout 300 python3 tools/exact_trace.py mm_QPILOTNO 2>/dev/null | cut -c1-220 > $O/r06_exact_engine_timeline.txt; tail -30 $O/r06_exact_engine_timeline.txt
# round 6, late: the factor-only benchmark through the class objects, the batched kernel's in-kernel split per batch size, PMC passes of the dense step and the C4 batch
timeout 900 python3 tools/dense_cholesky_factorization_benchmark.py > $O/r06_dense_cholesky_factorization_benchmark.txt 2>/dev/null; tail -4 $O/r06_dense_cholesky_factorization_benchmark.txt | cut -c1-200
timeout 300 python3 tools/prof_batch_split.py 1 256 1024 2048 4096 8192 16384 > $O/r06_batch_split.txt 2>&1
timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $O/pmc_f -- python3 tools/prof_dense.py 4096 4096 0 3 0 > /dev/null 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $O/pmc_w -- python3 tools/prof_dense.py 4096 4096 0 3 0 > /dev/null 2>&1
python3 tools/make_pmc_json.py $O/pmc_f $O/pmc_w 4096 4096 0 "round 6, final code" > $O/r06_pmc_dense_c2.json
timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $O/pmc_bf -- python3 tools/dbg_batch.py 8192 0 > /dev/null 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $O/pmc_bw -- python3 tools/dbg_batch.py 8192 0 > /dev/null 2>&1
python3 tools/rocprof_pmc.py $O/pmc_bf k_batch_ipm; python3 tools/rocprof_pmc.py $O/pmc_bw k_batch_ipm   # -> profiles/r06_pmc_batch_c4.json
rm -rf $O/pmc_f $O/pmc_w $O/pmc_bf $O/pmc_bw
