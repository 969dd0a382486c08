cd $GRAFT_REPO_ROOT
ulimit -v 150000000
mkdir -p gpurun_out
timeout 1200 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29911 bench.py --gpus 2 --steps 5 --warmup 2 --no-size-sweep --no-cpu-baseline > gpurun_out/r04_bench_n2_shared_gpu_gloo.json 2> gpurun_out/r04_bench_n2.err
echo "rc=$?"; tail -3 gpurun_out/r04_bench_n2.err | cut -c1-300
python3 -c "
import json
d=json.loads([l for l in open('gpurun_out/r04_bench_n2_shared_gpu_gloo.json') if l.startswith('{')][-1])
print(d['n_gpus'], d['value'], d['ms_per_step'], d['scaling'], d['process_group']['collective_backend'])
print({k:(v.get('value') if isinstance(v,dict) else v) for k,v in d.get('sparse_kkt',{}).items()})
c=d.get('stage_partitioned_c5',{}); print({k:c.get(k) for k in ('ms_per_step','single_gpu_ms_per_step','bitwise_equal_all_ranks','sharded_assembly','collective_backend')}); print(str(c.get('native_rccl_transport'))[:300])
print(d.get('batched_qp',{}).get('strong',{}).get('qp_per_s'))
"
