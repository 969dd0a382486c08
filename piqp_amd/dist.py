"""Multi-GPU plumbing for the batched / replica modes (torch.distributed over RCCL; gloo on CPU in tests).

The KKT hot path has no data-path collective: independent QP instances (or replicas of one shape) are partitioned
across ranks, every rank works on its contiguous shard, and the only communication is the barrier around the timed
region, the MAX-reduction of the elapsed time and the final gather of per-instance statistics."""
import os


def init(backend=None):
    """returns (rank, world, local_rank); initialises torch.distributed when WORLD_SIZE > 1"""
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        import torch
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if not dist.is_initialized():
            if backend is None:
                backend = "nccl" if torch.cuda.is_available() else "gloo"
            kw = {}
            if backend == "nccl":
                kw["device_id"] = torch.device("cuda", local_rank)
            dist.init_process_group(backend=backend, **kw)
    return rank, world, local_rank


def shard_range(total, rank, world):
    """contiguous, balanced partition of `total` independent units: sizes differ by at most one"""
    base, rem = divmod(total, world)
    start = rank * base + min(rank, rem)
    return start, start + base + (1 if rank < rem else 0)


def barrier():
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        dist.barrier()


def max_over_ranks(value, device=None):
    """MAX all-reduce of one python float (the elapsed time of the timed region)"""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=device if device is not None else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def gather_stats(local_rows, device=None):
    """all-gather of per-instance statistics rows (list of equal-length float lists) -> list over all instances in
    global instance order (shards are contiguous, so concatenation by rank is the global order)"""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return [list(r) for r in local_rows]
    world = dist.get_world_size()
    counts = [None] * world
    dist.all_gather_object(counts, len(local_rows))
    width = len(local_rows[0]) if local_rows else 0
    widths = [None] * world
    dist.all_gather_object(widths, width)
    width = max(widths)
    mx = max(counts)
    buf = torch.zeros((mx, max(width, 1)), dtype=torch.float64, device=device if device is not None else "cpu")
    if local_rows:
        buf[: len(local_rows), :width] = torch.tensor(local_rows, dtype=torch.float64)
    out = [torch.zeros_like(buf) for _ in range(world)]
    dist.all_gather(out, buf)
    rows = []
    for r in range(world):
        rows.extend(out[r][: counts[r], :width].cpu().tolist())
    return rows


def finalize():
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()
