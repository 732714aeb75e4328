"""The exchange plan of a scene group (rtx_gather_plan): which member moves which bytes where.  The group executes
exactly this list with RCCL, so checking it here covers the N > 1 bookkeeping that a one-GPU box cannot run."""
import numpy as np
import pytest

import ray_tracer_amd as rt


@pytest.mark.parametrize("n", [1, 2, 3, 8])
def test_gather_to_root_fills_the_roots_buffer_exactly_once(n):
    local = 3 * 32 * 32 * 127
    ops = rt.gather_plan(n, local, rt.GATHER_ROOT)
    covered = np.zeros(n * local, dtype=np.int32)
    sends, recvs = {}, {}
    for op in ops:
        assert op.bytes == local and op.src_offset == 0
        if op.kind == rt.GATHER_OP_SEND:
            assert op.member != 0 and op.peer == 0 and op.member not in sends
            sends[op.member] = op
        elif op.kind == rt.GATHER_OP_RECV:
            assert op.member == 0 and op.peer not in recvs
            recvs[op.peer] = op
            assert op.dst_offset == op.peer * local          # rank-major, what rtx_untile reads
            covered[op.dst_offset:op.dst_offset + op.bytes] += 1
        else:
            assert op.kind == rt.GATHER_OP_COPY and op.member == 0 and op.dst_offset == 0
            covered[:local] += 1
    assert sorted(sends) == sorted(recvs) == list(range(1, n))     # every send meets its receive
    assert (covered == 1).all()
    # a send and its receive are issued next to each other, receive first on the root: same order on both ends
    kinds = [(op.kind, op.member, op.peer) for op in ops if op.kind != rt.GATHER_OP_COPY]
    for r in range(1, n):
        assert kinds.index((rt.GATHER_OP_RECV, 0, r)) + 1 == kinds.index((rt.GATHER_OP_SEND, r, 0))


@pytest.mark.parametrize("n", [1, 2, 8])
def test_all_gather_plan_is_one_collective_per_member(n):
    ops = rt.gather_plan(n, 4096, rt.GATHER_ALL)
    assert [(op.kind, op.member, op.bytes) for op in ops] == [(rt.GATHER_OP_ALLGATHER, i, 4096) for i in range(n)]


def test_bad_plans_are_refused():
    with pytest.raises(rt.RtxError):
        rt.gather_plan(0, 16, rt.GATHER_ROOT)
    with pytest.raises(rt.RtxError):
        rt.gather_plan(2, 16, 7)
