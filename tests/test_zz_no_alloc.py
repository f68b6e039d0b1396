"""mpc_hip_run_seeded_static -- the entry point that allocates nothing (include/mpc_hip.h) -- against
mpc_hip_run_seeded, through the C-ABI, bit for bit: every candidate's costs, the packed winners, the fetched result
and its trajectory.  Two calls in a row on the same handle, the second with a smaller input (the fixed device array
keeps the first call's contents), one problem and a small fleet, both precisions; sizes beyond the array's capacity
are rejected before anything is launched.  Collected last on purpose (the round's end runs pytest -x)."""
import numpy as np
import pytest

import mpc_amd as M
from mpc_amd import workloads as W
from helpers import traj_array

pytestmark = pytest.mark.gpu


def _torch():
    import torch
    assert torch.cuda.is_available()
    return torch


def _run(solver, static, n_problems, first, count, prec, selection):
    torch = _torch()
    last = torch.full((n_problems * count,), -1.0, dtype=torch.float64, device="cuda")
    cum = torch.full((n_problems * count,), -1.0, dtype=torch.float64, device="cuda")
    win = torch.zeros(2 * n_problems, dtype=torch.int64, device="cuda")
    run = solver.run_seeded_static if static else solver.run_seeded
    run(first, count, prec, selection, last, cum)
    solver.copy_winners(win)
    res = solver.fetch()
    return last.cpu().numpy(), cum.cpu().numpy(), win.cpu().numpy(), res


def _same(a, b, n_problems):
    la, ca, wa, ra = a
    lb, cb, wb, rb = b
    assert np.array_equal(la.view(np.int64), lb.view(np.int64))           # bit-exact costs of every candidate
    assert np.array_equal(ca.view(np.int64), cb.view(np.int64))
    assert np.array_equal(wa, wb)                                         # (score bits, ordinal) per problem
    for p in range(n_problems):
        x, y = ra[p], rb[p]
        assert (x.status, x.levels, x.best_ordinal, x.evaluated, x.rollout_steps) == \
               (y.status, y.levels, y.best_ordinal, y.evaluated, y.rollout_steps)
        assert (x.best_last, x.best_cumulative, x.first_rudder, x.first_throttle) == \
               (y.best_last, y.best_cumulative, y.first_rudder, y.first_throttle)
        assert np.array_equal(traj_array(x).view(np.int64), traj_array(y).view(np.int64))


@pytest.mark.parametrize("prec", [M.PREC_FAST, M.PREC_F64])
def test_static_entry_point_equals_run_seeded_twice_in_a_row(prec):
    solver = M.Solver(0)                  # its own handle: the first static call of this handle comes first
    try:
        p = W.config2_problem(depth=12)
        solver.upload([p], [W.SEED])
        # 1st call: 65,536 candidates from ordinal 1000; 2nd call, same handle, straight after: 5,000 from ordinal 7
        big_s = _run(solver, True, 1, 1000, 65536, prec, M.SELECT_LAST_STEP)
        small_s = _run(solver, True, 1, 7, 5000, prec, M.SELECT_LAST_STEP)
        big = _run(solver, False, 1, 1000, 65536, prec, M.SELECT_LAST_STEP)
        small = _run(solver, False, 1, 7, 5000, prec, M.SELECT_LAST_STEP)
        _same(big_s, big, 1)
        _same(small_s, small, 1)
        assert big[3][0].status == 0 and 1000 <= big[3][0].best_ordinal < 1000 + 65536
        assert small[3][0].status == 0 and 7 <= small[3][0].best_ordinal < 7 + 5000
        # cumulative selection goes through the same array
        _same(_run(solver, True, 1, 0, 3001, prec, M.SELECT_CUMULATIVE), _run(solver, False, 1, 0, 3001, prec, M.SELECT_CUMULATIVE), 1)
    finally:
        solver.close()


def test_static_entry_point_on_a_fleet_then_a_smaller_one():
    solver = M.Solver(0)
    try:
        probs, seeds = W.fleet_problems(48, depth=10)
        solver.upload(probs, seeds)
        _same(_run(solver, True, 48, 0, 2048, M.PREC_FAST, M.SELECT_LAST_STEP),
              _run(solver, False, 48, 0, 2048, M.PREC_FAST, M.SELECT_LAST_STEP), 48)
        solver.upload(probs[:5], seeds[:5])
        _same(_run(solver, True, 5, 0, 700, M.PREC_FAST, M.SELECT_LAST_STEP),
              _run(solver, False, 5, 0, 700, M.PREC_FAST, M.SELECT_LAST_STEP), 5)
    finally:
        solver.close()


def test_static_entry_point_rejects_what_does_not_fit():
    solver = M.Solver(0)
    try:
        p = W.config2_problem(depth=4)
        solver.upload([p], [W.SEED])
        with pytest.raises(M.MpcError) as e:
            solver.run_seeded_static(0, (1 << 20) + 1)
        assert e.value.status == 8 and "MPC_STATIC_MAX_CANDIDATES" in str(e.value)
        solver.run_seeded_static(0, 1 << 20)                     # the maximum itself runs (256-lane workgroups: 4096 of them)
        r = solver.fetch()[0]
        solver.run_seeded(0, 1 << 20)
        assert (r.best_ordinal, r.best_last) == (solver.fetch()[0].best_ordinal, solver.fetch()[0].best_last)
        solver.upload([p] * 33000, [W.SEED] * 33000)             # 33,000 problems x 1 workgroup each > 2^15 workgroups
        with pytest.raises(M.MpcError) as e:
            solver.run_seeded_static(0, 2)
        assert e.value.status == 8 and "MPC_STATIC_MAX_WORKGROUPS" in str(e.value)
        solver.upload([p] * 32768, [W.SEED] * 32768)             # exactly 2^15 fits
        solver.run_seeded_static(0, 2)
        a = [(r.best_ordinal, r.best_last) for r in solver.fetch()]
        solver.run_seeded(0, 2)
        assert a == [(r.best_ordinal, r.best_last) for r in solver.fetch()] and len(set(a)) == 1
    finally:
        solver.close()


def test_static_entry_point_allocates_nothing():
    """The property the entry point is named for: device memory in use (hipMemGetInfo, through torch.cuda.mem_get_info)
    does not move across the FIRST static call of a fresh handle -- a call larger than anything that handle has run,
    which would make mpc_hip_run_seeded grow its keys / triples / snapshot buffers -- nor across a second, different one.
    (The kernels' code objects are loaded beforehand by another handle: loading code is not what is being measured.)"""
    torch = _torch()
    p = W.config2_problem(depth=12)
    warm = M.Solver(0)
    warm.upload([p], [W.SEED])
    warm.run_seeded_static(0, 70000, M.PREC_FAST)
    warm.run_seeded_static(0, 70000, M.PREC_F64)
    warm.fetch()
    solver = M.Solver(0)
    try:
        solver.upload([p], [W.SEED])                             # allocates (documented): the problems must be resident
        solver.synchronize()
        torch.cuda.synchronize()
        free0 = torch.cuda.mem_get_info()[0]
        solver.run_seeded_static(0, 1 << 20, M.PREC_FAST)        # 2^20 candidates: 8 MB of keys if it were to allocate them
        solver.synchronize()
        free1 = torch.cuda.mem_get_info()[0]
        solver.run_seeded_static(5, 300000, M.PREC_F64)
        solver.synchronize()
        free2 = torch.cuda.mem_get_info()[0]
        assert free1 == free0 and free2 == free0, (free0, free1, free2)
        # the allocating entry point, for contrast, does grow its buffers on such a call (the check can see an allocation)
        solver.run_seeded(0, 1 << 20, M.PREC_FAST)
        solver.synchronize()
        assert torch.cuda.mem_get_info()[0] < free0
    finally:
        solver.close()
        warm.close()
