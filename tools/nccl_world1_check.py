#!/usr/bin/env python3
"""The RCCL code path of piqp_amd.dist with a one-rank "nccl" group (all a 1-GPU box can exercise): bookkeeping collectives on device
tensors and the two collectives StagePartition issues.  Prints OK."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.update(RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=os.environ.get("MASTER_PORT", "29599"))
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

torch.cuda.set_device(0)
dist.init_process_group(backend="nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
from piqp_amd import dist as pd  # noqa: E402

assert pd.max_over_ranks(1.25) == 1.25
rows = pd.gather_stats([[1.0, 2.0], [3.0, 4.0]])
assert rows == [[1.0, 2.0], [3.0, 4.0]], rows
pd.barrier()
t = torch.arange(8, dtype=torch.float64, device="cuda")
dist.all_reduce(t)
buf = torch.zeros(8, dtype=torch.float64, device="cuda")
dist.all_gather_into_tensor(buf, t.clone())
torch.cuda.current_stream().synchronize()
assert torch.equal(buf, t)
dist.destroy_process_group()
print("OK")
