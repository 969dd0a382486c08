"""world_size-2 gloo tests (CPU) of the multi-GPU plumbing used by bench.py: sharding of independent units,
barrier, MAX-over-ranks timing reduction and the final statistics gather."""
import os
import socket

import pytest
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from piqp_amd import dist as pd
    r, w, lr = pd.init(backend="gloo")
    lo, hi = pd.shard_range(11, r, w)
    rows = [[float(i), float(i * i)] for i in range(lo, hi)]
    pd.barrier()
    elapsed = pd.max_over_ranks(1.0 + r)
    allrows = pd.gather_stats(rows)
    q.put((r, lo, hi, elapsed, allrows))
    pd.finalize()


def test_shard_range_partitions_exactly():
    from piqp_amd.dist import shard_range
    for total in (0, 1, 7, 8, 8192, 8193):
        for world in (1, 2, 3, 8):
            pieces = [shard_range(total, r, world) for r in range(world)]
            assert pieces[0][0] == 0 and pieces[-1][1] == total
            assert all(pieces[i][1] == pieces[i + 1][0] for i in range(world - 1))
            sizes = [b - a for a, b in pieces]
            assert max(sizes) - min(sizes) <= 1


def test_two_rank_gloo_roundtrip():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(2)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    res.sort()
    assert (res[0][1], res[0][2], res[1][1], res[1][2]) == (0, 6, 6, 11)
    for r in res:
        assert r[3] == 2.0  # MAX over ranks of (1 + rank)
        assert r[4] == [[float(i), float(i * i)] for i in range(11)]  # global instance order
