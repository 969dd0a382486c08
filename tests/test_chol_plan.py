"""CPU test of the persistent dense factorisation's ticket list (csrc/dense_kernels.hip: chol_build_tasks / chol_build_tasks_fused), fetched through the host-only
C-ABI call pq_debug_chol_plan.  The launch is deadlock-free under ANY number of resident workgroups because a workgroup inside a task only ever waits for results
of EARLIER tickets; this test replays the list in ticket order with every earlier task complete and checks, for each task, exactly the conditions the kernel
waits for (k_chol_persistent) -- plus coverage: every tile receives its assembly slices and the updates of panels 0 .. j-1 once, in order (the factorisation of
Eigen::LLT / dense/ldlt_no_pivot.hpp:313-354 as a task graph)."""
import ctypes as C

import numpy as np
import pytest

from piqp_amd import _lib

FAST_ROWS = 2


def split(T, k):
    return 1 if T - k - 1 >= 22 else 2


def split_row(T, k, ti):
    return 2 if ti <= FAST_ROWS else split(T, k)


def asm_slices(j, mch):
    s = 8 if j <= 2 else (4 if j <= 8 else (2 if j <= 16 else 1))
    return s if s < mch else max(mch, 1)


def plan(T, mch):
    L = _lib.load()
    n = L.pq_debug_chol_plan(T, mch, None, 0)
    assert n > 0
    out = np.zeros((n, 6), dtype=np.int32)
    assert L.pq_debug_chol_plan(T, mch, out.ctypes.data_as(C.c_void_p), n) == n
    return out


@pytest.mark.parametrize("T,mch", [(3, 0), (8, 0), (32, 0), (3, 4), (5, 8), (8, 32), (16, 16), (32, 32), (32, 5), (40, 64)])
def test_every_task_waits_for_earlier_tickets_only(T, mch):
    tasks = plan(T, mch)
    fused = mch > 0
    tver = -np.ones((T, T), dtype=int) if fused else np.zeros((T, T), dtype=int)  # updates received; -1: not assembled yet
    if fused:
        tver[:, 0] = 0
    acnt = np.zeros((T, T), dtype=int)
    lready = np.zeros((T + 1, T), dtype=int)
    pdone = np.zeros(T, dtype=int)
    owner_seen = np.zeros(T, dtype=bool)
    dhalf = np.zeros(T, dtype=int)
    panel_final = np.zeros((T, T), dtype=bool)
    slices_seen = {}

    def rows_ready(k, i):
        return k == 0 or lready[k, i] == split_row(T, k - 1, i - k)

    def rounds_complete():
        p = 0
        while p + 1 < T and pdone[p] == sum(split_row(T, p, ti) for ti in range(1, T - p - 1)):
            p += 1
        return p

    # fused assembly: the assembly tasks (kind 6) sit in eight queues behind the ticket list (gate = queue id), the list holds one token (kind 8) per task.  A task
    # that finds its tile not assembled runs assembly tasks itself until it is (help_assembled in the kernel), so for the replay every tile counts as assembled;
    # what is checked is that the queues hold every slice of every tile exactly once, tile (i, j) in queue i mod 8, and that there are as many tokens as tasks.
    queue = tasks[tasks[:, 0] == 6]
    tasks = tasks[tasks[:, 0] != 6]
    if fused:
        assert (tasks[:, 0] == 8).sum() == len(queue)
        assert (np.diff(queue[:, 4]) >= 0).all()  # queue after queue
        for kind, sl, i, j, qid, aux in queue:
            S = asm_slices(j, mch)
            assert 1 <= j <= i < T and 0 <= sl < S and qid == (i & 7) and (i, j, sl) not in slices_seen
            slices_seen[(i, j, sl)] = aux
            acnt[i, j] += 1
        tver[tver < 0] = 0
    else:
        assert len(queue) == 0

    for pos, (kind, rnd, a, b, gate, aux) in enumerate(tasks):
        assert gate <= rounds_complete(), (pos, kind, rnd, a, b, gate)  # the gate only ever refers to rounds finished by earlier tickets
        if kind == 8:
            assert fused
            continue
        if kind == 7:
            i, j, klo, khi = a, b, rnd, gate + 1
            assert 1 <= j <= i < T and 0 <= klo < khi <= j - 3
            assert tver[i, j] == klo, (pos, i, j, klo, tver[i, j])
            assert rows_ready(khi - 1, i) and rows_ready(khi - 1, j), (pos, i, j, khi)
            tver[i, j] = khi
            continue
        k = rnd
        d = k + 1
        if kind in (0, 1):
            assert tver[d, d] == k, (pos, kind, k, tver[d, d])
            assert rows_ready(k, d)
            if kind == 1:
                if k >= 2:
                    assert pdone[k - 2] == sum(split_row(T, k - 2, ti) for ti in range(1, T - (k - 2) - 1))
                owner_seen[k] = True
            continue
        ti = a
        tj = 0 if kind == 2 else b
        i, j = k + 1 + ti, k + 1 + tj
        assert tver[i, j] == k, (pos, kind, k, i, j, tver[i, j])
        assert rows_ready(k, i) and rows_ready(k, j), (pos, kind, k, i, j)
        if kind == 2:
            assert owner_seen[k], (pos, k)  # the panel rows follow the factorisation of the diagonal block
            lready[k + 1, i] += 1
            pdone[k] += 1
            if lready[k + 1, i] == split_row(T, k, ti):
                panel_final[i, j] = True
        elif kind in (4, 5):
            dhalf[k] += 1
            if dhalf[k] == 2:
                tver[i, j] = k + 1
        else:
            tver[i, j] = k + 1
    # coverage
    for j in range(1, T):
        for i in range(j, T):
            assert tver[i, j] == j - 1, (i, j, tver[i, j])  # (the last update, k = j - 1, is the crew's on the diagonal tile and the panel task's below it)
            if i > j:
                assert panel_final[i, j], (i, j)
            if fused:
                assert acnt[i, j] == asm_slices(j, mch)
    assert owner_seen[: T - 1].all()
    if fused:
        # the partial-sum slots of the K-sliced tiles do not overlap
        used = set()
        for (i, j, sl), aux in slices_seen.items():
            if asm_slices(j, mch) > 1:
                assert (aux + sl) not in used
                used.add(aux + sl)
