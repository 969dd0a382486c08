cd $GRAFT_REPO_ROOT
ulimit -v 100000000
mkdir -p gpurun_out
for w in 2 4; do
timeout 600 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node=$w --master-addr 127.0.0.1 --master-port $((29800+w)) tools/dist_c5.py --stages 5000 --steps 5 --warmup 2 --backend multistage 2>/dev/null | grep "^{" | tail -1 > gpurun_out/r04_sharded_assembly_w$w.json
python3 -c "
import json; d=json.load(open('gpurun_out/r04_sharded_assembly_w$w.json')); print($w, d['bitwise_equal_all_ranks'], d['sharded_assembly'], d.get('sharded_assembly_entries_single_gpu'), d['ms_per_step'], d['single_gpu_ms_per_step'])"
done
