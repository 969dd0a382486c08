"""Multi-GPU plumbing for the batched / replica modes (torch.distributed over RCCL; gloo on CPU in tests).

The KKT hot path has no data-path collective: independent QP instances (or replicas of one shape) are partitioned
across ranks, every rank works on its contiguous shard, and the only communication is the barrier around the timed
region, the MAX-reduction of the elapsed time and the final gather of per-instance statistics."""
import os


def init(backend=None):
    """returns (rank, world, device_index); initialises torch.distributed when WORLD_SIZE > 1.

    One rank per GPU over RCCL ("nccl").  When the node has fewer GPUs than ranks (the 1-GPU test boxes) the ranks share
    the devices round-robin and the process group is "gloo": the same code path end to end, host-staged collectives."""
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dev_index = local_rank
    import torch
    if torch.cuda.is_available():
        ngpu = torch.cuda.device_count()
        dev_index = local_rank % max(ngpu, 1)
        if backend is None and (world > 1 or os.environ.get("PIQP_AMD_FORCE_PG")):
            backend = "nccl" if ngpu >= int(os.environ.get("LOCAL_WORLD_SIZE", world)) else "gloo"
        torch.cuda.set_device(dev_index)
    if world > 1 or os.environ.get("PIQP_AMD_FORCE_PG"):  # PIQP_AMD_FORCE_PG=1: a one-rank group (exercises the RCCL path on a 1-GPU box)
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29598")
        os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
        if not dist.is_initialized():
            if backend is None:
                backend = "gloo"
            kw = {}
            if backend == "nccl":
                kw["device_id"] = torch.device("cuda", dev_index)
            dist.init_process_group(backend=backend, **kw)
    return rank, world, dev_index


def _coll_device(device):
    """tensors of the bookkeeping collectives: on this rank's GPU with RCCL ("nccl" has no CPU path), on the host otherwise"""
    import torch
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        if dist.get_backend() == "nccl":
            return device if device is not None else torch.device("cuda", torch.cuda.current_device())
        return "cpu"
    return device if device is not None else "cpu"


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
    t = torch.tensor([float(value)], dtype=torch.float64, device=_coll_device(device))
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
    buf = torch.zeros((mx, max(width, 1)), dtype=torch.float64, device=_coll_device(device))
    if local_rows:
        buf[: len(local_rows), :width] = torch.tensor(local_rows, dtype=torch.float64)
    out = [torch.zeros_like(buf) for _ in range(world)]
    dist.all_gather(out, buf)
    rows = []
    for r in range(world):
        rows.extend(out[r][: counts[r], :width].cpu().tolist())
    return rows


class StagePartition:
    """Stage-partitioned execution of ONE KKT system / solver over the ranks of a process group (BASELINE configs[4]).

    Every rank builds the same object on the same data and wraps it here.  The C library splits the assembly-tree work of
    factor / solve (pq_kkt_partition, include/piqp_amd.h) and calls back whenever data has to cross ranks; the collectives
    themselves are torch.distributed calls on device tensors -- RCCL over xGMI with the "nccl" backend, host-staged with
    "gloo" (the CPU-rendezvous tests, where several ranks share one GPU)."""

    def __init__(self, obj, rank=None, world=None, group=None, native=None):
        import ctypes as C

        import torch
        import torch.distributed as dist

        from . import _lib
        self.torch, self.dist, self.group = torch, dist, group
        L = _lib.load()
        on = dist.is_available() and dist.is_initialized()
        self.rank = rank if rank is not None else (dist.get_rank(group) if on else 0)
        self.world = world if world is not None else (dist.get_world_size(group) if on else 1)
        self.backend = dist.get_backend(group) if on else None
        is_solver = hasattr(obj, "solve") and hasattr(obj, "setup")
        h = obj.h if is_solver else (obj.backend().h if hasattr(obj, "backend") else obj.h)
        part = L.pq_solver_partition if is_solver else L.pq_kkt_partition
        setx = L.pq_solver_set_exchange if is_solver else L.pq_kkt_set_exchange
        sizes = (C.c_longlong * 3)()
        _lib.check(part(h, self.rank, self.world, sizes), "partition")
        self.sizes = [int(v) for v in sizes]
        dev = torch.device("cuda", torch.cuda.current_device())
        self.buf_factor = torch.zeros(self.sizes[0], dtype=torch.float64, device=dev)
        self.buf_forward = torch.zeros(self.sizes[1], dtype=torch.float64, device=dev)
        self.buf_gather = torch.zeros(self.world * self.sizes[2], dtype=torch.float64, device=dev)
        self.buf_norm = torch.zeros(2, dtype=torch.float64, device=dev)  # which = 3: ||err||_inf of the sharded refinement residual
        self.calls = [0, 0, 0, 0]
        # native = True: the library's own RCCL transport (pq_kkt_set_comm_rccl: collectives enqueued on the handle's stream, no callback).
        # Default: native whenever the process group runs on RCCL ("nccl"), i.e. one GPU per rank; the callback path below remains for gloo
        # (several ranks sharing one GPU in the CPU-rendezvous tests) and as the reference implementation of the protocol.
        # Round 3: no multi-rank run of the native transport exists yet (no multi-GPU node was available to the build), so it is OPT-IN
        # (native=True or PIQP_AMD_NATIVE_RCCL=1) and the callback transport -- torch.distributed's own RCCL collectives on the registered device
        # buffers -- is the default; bench.py --gpus N runs both and compares them bit for bit.
        self.native = (self.backend == "nccl" and os.environ.get("PIQP_AMD_NATIVE_RCCL") == "1"
                       and os.environ.get("PIQP_AMD_CALLBACK_EXCHANGE") is None) if native is None else bool(native)
        if self.native and self.world > 1 and not on:
            raise RuntimeError("StagePartition(native=True) with world > 1 needs an initialised torch.distributed process group to ship the RCCL unique id")
        self.error = None
        self._cb = _lib.EXCHANGE_FN(self._exchange)  # must outlive the handle's use of it
        self._obj = obj
        if self.native:
            idb = (C.c_ubyte * 128)()
            if self.rank == 0:
                _lib.check(L.pq_rccl_unique_id(idb), "rccl_unique_id")
            box = [bytes(idb)]
            if on and self.world > 1:
                # `src` is a GLOBAL rank; the id comes from the group's rank 0
                src = dist.get_global_rank(group, 0) if group is not None else 0
                dist.broadcast_object_list(box, src=src, group=group, device=dev if self.backend == "nccl" else None)
            idb = (C.c_ubyte * 128).from_buffer_copy(box[0])
            setc = L.pq_solver_set_comm_rccl if is_solver else L.pq_kkt_set_comm_rccl
            _lib.check(setc(h, idb, self.rank, self.world), "set_comm_rccl")
        else:
            _lib.check(setx(h, self._cb, None, self.buf_factor.data_ptr(), self.buf_forward.data_ptr(), self.buf_gather.data_ptr()), "set_exchange")
            if self.world > 1 and os.environ.get("PIQP_AMD_REPLICATED_RESIDUAL") is None:
                # the sharded refinement residual (KKT_FULL backends; the others ignore the buffer): one all-reduce(MAX) per refinement step
                (L.pq_solver_set_exchange_norm if is_solver else L.pq_kkt_set_exchange_norm)(h, self.buf_norm.data_ptr())
        torch.cuda.synchronize(dev)
        self.dev = dev

    def _exchange(self, user, which):
        # called from inside pq_kkt_update_scalings_and_factor / pq_kkt_solve with the handle's stream drained
        try:
            torch, dist = self.torch, self.dist
            self.calls[which] += 1
            if self.world == 1 and not (dist.is_available() and dist.is_initialized()):
                return 0
            staged = self.backend != "nccl"
            if which == 3:
                t = self.buf_norm
                if staged:
                    c = t.cpu()
                    dist.all_reduce(c, op=dist.ReduceOp.MAX, group=self.group)
                    t.copy_(c)
                else:
                    dist.all_reduce(t, op=dist.ReduceOp.MAX, group=self.group)
            elif which in (0, 1):
                t = self.buf_factor if which == 0 else self.buf_forward
                if staged:
                    c = t.cpu()
                    dist.all_reduce(c, group=self.group)
                    t.copy_(c)
                else:
                    dist.all_reduce(t, group=self.group)
            else:
                sz = self.sizes[2]
                mine = self.buf_gather[self.rank * sz:(self.rank + 1) * sz]
                if staged:
                    c = mine.cpu()
                    outs = [torch.empty_like(c) for _ in range(self.world)]
                    dist.all_gather(outs, c, group=self.group)
                    self.buf_gather.copy_(torch.cat(outs))
                else:
                    dist.all_gather_into_tensor(self.buf_gather, mine.clone(), group=self.group)
            torch.cuda.current_stream(self.dev).synchronize()
            return 0
        except Exception as e:  # never let an exception cross the C frame
            self.error = e
            return 1

    def exchange_calls(self):
        """collectives performed so far, [which = 0, 1, 2]: counted by the callback, or read from the library for the native transport"""
        if not self.native:
            return list(self.calls[:3])
        import ctypes as C

        from . import _lib
        L = _lib.load()
        obj = self._obj
        is_solver = hasattr(obj, "solve") and hasattr(obj, "setup")
        h = obj.h if is_solver else (obj.backend().h if hasattr(obj, "backend") else obj.h)
        out = (C.c_int * 3)()
        _lib.check((L.pq_solver_native_exchange_calls if is_solver else L.pq_kkt_native_exchange_calls)(h, out), "native_exchange_calls")
        return [int(v) for v in out]

    def sharded_calls(self):
        """[sharded refinement-residual evaluations so far, rows of the KKT system in this rank's share] (pq_kkt_sharded_calls)"""
        import ctypes as C

        from . import _lib
        L = _lib.load()
        obj = self._obj
        is_solver = hasattr(obj, "solve") and hasattr(obj, "setup")
        h = obj.h if is_solver else (obj.backend().h if hasattr(obj, "backend") else obj.h)
        out = (C.c_int * 2)()
        _lib.check((L.pq_solver_sharded_calls if is_solver else L.pq_kkt_sharded_calls)(h, C.byref(out)), "sharded_calls")
        return [int(out[0]), int(out[1])]

    def sharded_solve_calls(self):
        """[sharded residual evaluations, residual rows in this rank's share, backend solves that folded / recovered on this rank's rows only, gathers of the
        eliminated multipliers, x rows folded per such solve, constraint rows recovered per such solve] (pq_kkt_sharded_solve_calls)"""
        import ctypes as C

        from . import _lib
        L = _lib.load()
        obj = self._obj
        is_solver = hasattr(obj, "solve") and hasattr(obj, "setup")
        h = obj.h if is_solver else (obj.backend().h if hasattr(obj, "backend") else obj.h)
        out = (C.c_int * 6)()
        _lib.check((L.pq_solver_sharded_solve_calls if is_solver else L.pq_kkt_sharded_solve_calls)(h, C.byref(out)), "sharded_solve_calls")
        return [int(v) for v in out]

    def comm_info(self):
        """what ran the collectives, as seen from the inside (the figures a multi-GPU bench line carries so that "RCCL saw N ranks" can be checked):
        transport, the process group's backend and size, and for the native transport the library communicator's own ncclCommCount / rank / device"""
        import ctypes as C

        from . import _lib
        L = _lib.load()
        obj = self._obj
        is_solver = hasattr(obj, "solve") and hasattr(obj, "setup")
        h = obj.h if is_solver else (obj.backend().h if hasattr(obj, "backend") else obj.h)
        out = (C.c_int * 4)()
        _lib.check((L.pq_solver_comm_info if is_solver else L.pq_kkt_comm_info)(h, out), "comm_info")
        on = self.dist.is_available() and self.dist.is_initialized()
        return dict(transport=("none", "callback", "native")[out[0]], process_group_backend=self.backend, process_group_size=(self.dist.get_world_size(self.group) if on else 1),
                    library_comm_count=int(out[1]), library_comm_rank=int(out[2]), library_comm_device=int(out[3]), device=int(self.dev.index),
                    exchange_bytes=[8 * int(self.sizes[0]), 8 * int(self.sizes[1]), 8 * int(self.sizes[2]) * int(self.world)])

    def info(self):
        import ctypes as C

        from . import _lib
        L = _lib.load()
        obj = self._obj
        if hasattr(obj, "solve") and hasattr(obj, "setup"):
            return None
        h = obj.backend().h if hasattr(obj, "backend") else obj.h
        out = (C.c_int * 8)()
        _lib.check(L.pq_kkt_partition_info(h, out), "partition_info")
        return dict(owned_supernodes=out[0], shared_supernodes=out[1], boundary_roots=out[2], span=(out[3], out[4]), work_permille=out[5],
                    shared_work_permille=out[6], world=out[7], exchange_doubles=self.sizes)


def spawn_waiting(argv, extra_env=None, port_offset=37):
    """Starts `python argv...` as a child that blocks on its stdin until release_and_collect() -- to be called BEFORE this process
    touches the GPU (a process that has initialised HIP must not fork + exec).  The child inherits RANK / WORLD_SIZE / LOCAL_RANK /
    MASTER_ADDR and gets its own MASTER_PORT, so the children of all ranks form a second, independent process group: a hang or a
    crash in there cannot take the parent's group (and its one JSON line) down."""
    import subprocess
    import sys
    env = dict(os.environ)
    env["MASTER_PORT"] = str(int(env.get("MASTER_PORT", "29500")) + port_offset)
    for k in ("TORCHELASTIC_RUN_ID", "TORCHELASTIC_USE_AGENT_STORE"):  # the children rendezvous on their own TCPStore (rank 0 hosts it), not on the launcher's
        env.pop(k, None)
    if extra_env:
        env.update(extra_env)
    return subprocess.Popen([sys.executable] + list(argv), stdin=subprocess.PIPE, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env, text=True)


def release_and_collect(proc, timeout):
    """lets the waiting child run, returns (returncode or None on timeout, stdout, stderr tail); kills exactly that child on timeout"""
    import subprocess
    try:
        out, err = proc.communicate(input="go\n", timeout=timeout)
        return proc.returncode, out, err[-2000:]
    except subprocess.TimeoutExpired:
        proc.kill()
        out, err = proc.communicate()
        return None, out, (err or "")[-2000:]


def finalize():
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()
