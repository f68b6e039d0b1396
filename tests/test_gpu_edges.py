"""GPU parity at the edges of the C-ABI's domain: maximum depth and substep count, deep explicit control arrays
(two staging chunks), sets of one candidate, candidates without a finite score, sets in which everything ties."""
import math

import numpy as np
import pytest

import mpc_amd as M
from mpc_amd import binding as B
from mpc_amd import workloads as W
from helpers import oracle_seeded, oracle_explicit, rel_err

pytestmark = pytest.mark.gpu

TOL_FAST = 1e-5
TOL_F64 = 1e-12


def _circle_problem(depth, substeps, current=(0.2, -0.1)):
    """A reference trajectory that needs no Dubins plan: a 40-m circle at 2 m/s sampled every second."""
    r, v = 40.0, 2.0
    cps = []
    for i in range(depth):
        th = v * (i + 1) / r
        cps.append((r * math.sin(th), r * (1.0 - math.cos(th)), (math.pi / 2 - th) % (2 * math.pi), v))
    return B.make_problem((0.5, -0.5, math.pi / 2 - 0.05, 1.8), cps, substeps, last=(0.1, 0.6), current=current)


def _torch():
    import torch
    assert torch.cuda.is_available(), "these tests need the GPU"
    return torch


@pytest.mark.parametrize("substeps", ["uniform64", "reference_native_64"])
def test_maximum_depth_and_substeps(solver, substeps):
    """depth = MPC_MAX_DEPTH levels of MPC_MAX_SUBSTEPS substeps: 4096 Euler steps per candidate."""
    torch = _torch()
    sub = [1.0 / 64] * 64 if substeps == "uniform64" else B.substep_schedule(6.35)
    assert len(sub) == B.MAX_SUBSTEPS
    p = _circle_problem(B.MAX_DEPTH, sub)
    n = 512
    o_last, o_cum, o_best = oracle_seeded(p, W.SEED, 0, n)
    assert np.isfinite(o_last).sum() > n // 2          # 6.35-s levels outrun the reference: some costs overflow, as on the CPU
    for prec, tol in [(M.PREC_F64, TOL_F64), (M.PREC_FAST, TOL_FAST)]:
        last = torch.empty(n, dtype=torch.float64, device="cuda")
        cum = torch.empty(n, dtype=torch.float64, device="cuda")
        res = solver.eval_seeded(p, W.SEED, 0, n, prec, M.SELECT_LAST_STEP, last, cum)
        assert rel_err(last.cpu().numpy(), o_last).max() <= tol, (substeps, prec)
        assert rel_err(cum.cpu().numpy(), o_cum).max() <= tol, (substeps, prec)
        assert res.best_ordinal == o_best.best_index and res.levels == B.MAX_DEPTH
        assert res.rollout_steps == n * B.MAX_DEPTH * B.MAX_SUBSTEPS


@pytest.mark.parametrize("depth", [33, 64])
def test_deep_explicit_control_arrays(solver, depth):
    """explicit FP32 controls deeper than one staging chunk (32 levels) and at the LDS cap (depth 64)."""
    torch = _torch()
    p = _circle_problem(depth, W.uniform_substeps())
    n = 300                                   # ragged: 4 full waves + 44 lanes
    rud, thr = W.seeded_controls(W.SEED + depth, 0, n, depth, p.last_rudder, p.last_throttle)
    rud = rud.astype(np.float32).astype(np.float64)          # the ABI takes FP32 controls (hold-last 0.1 / 0.6 is not FP32)
    thr = thr.astype(np.float32).astype(np.float64)
    o_last, o_cum, o_best = oracle_explicit(p, rud, thr)
    d_r = torch.tensor(rud, dtype=torch.float32, device="cuda").contiguous()
    d_t = torch.tensor(thr, dtype=torch.float32, device="cuda").contiguous()
    for prec, tol in [(M.PREC_F64, TOL_F64), (M.PREC_FAST, TOL_FAST)]:
        last = torch.empty(n, dtype=torch.float64, device="cuda")
        res = solver.eval_candidates(p, d_r, d_t, n, prec, M.SELECT_LAST_STEP, last, None)
        assert rel_err(last.cpu().numpy(), o_last).max() <= tol
        assert res.best_ordinal == o_best.best_index
        got = [(res.trajectory[l + 1].rudder, res.trajectory[l + 1].throttle) for l in range(depth)]
        want = [(float(rud[l, res.best_ordinal]), float(thr[l, res.best_ordinal])) for l in range(depth)]
        assert got == want                     # the winner's own controls, every level


def test_a_set_of_one_candidate(solver):
    p = W.config2_problem(depth=1)
    for prec, tol in [(M.PREC_F64, TOL_F64), (M.PREC_FAST, TOL_FAST)]:
        for first in (0, 1, 12345):
            o_last, _, _ = oracle_seeded(p, W.SEED, first, 1)
            res = solver.eval_seeded(p, W.SEED, first, 1, prec)
            assert res.best_ordinal == first and res.levels == 1 and res.evaluated == 1
            assert rel_err(res.best_last, o_last[0]) <= tol


def test_no_finite_score_selects_nothing(solver):
    """exp(distance) overflows for every candidate: nothing is ever below the reference's initial minScore =
    DBL_MAX (controller.cpp:81,160).  One problem: the call fails with the reference's exception text; in a batch the
    problem carries the status and the others are unaffected."""
    torch = _torch()
    far = W.config2_problem(depth=3)
    for i in range(far.depth):
        far.checkpoints[i].x += 5000.0           # exp(5000) = inf
    o_last, _, _ = oracle_seeded(far, W.SEED, 0, 256)
    assert np.isinf(o_last).all()
    for prec in (M.PREC_F64, M.PREC_FAST):
        last = torch.empty(256, dtype=torch.float64, device="cuda")
        with pytest.raises(M.MpcError) as e:
            solver.eval_seeded(far, W.SEED, 0, 256, prec, M.SELECT_LAST_STEP, last, None)
        assert e.value.status == 5 and "single iteration" in str(e.value)
        assert np.isinf(last.cpu().numpy()).all()
    near = W.config2_problem(depth=3)
    res = solver.eval_seeded_batch([near, far, near], [W.SEED] * 3, 256, M.PREC_FAST)
    assert [r.status for r in res] == [0, 5, 0]
    assert res[0].best_ordinal == res[2].best_ordinal and res[0].best_last == res[2].best_last
    assert res[1].best_ordinal == 2 ** 64 - 1 and math.isinf(res[1].best_last)
    # the combine step keeps "nothing selected" apart from real winners
    w = torch.empty(6, dtype=torch.int64, device="cuda")
    solver.copy_winners(w)
    w = w.cpu().numpy().view(np.uint64)
    assert w[2] == 0x7FF0000000000000 and w[3] == 2 ** 64 - 1


def test_sizes_beyond_the_maxima_are_refused(solver):
    p = W.config2_problem(depth=3)
    p.depth = B.MAX_DEPTH + 1
    with pytest.raises(M.MpcError) as e:
        solver.eval_seeded(p, W.SEED, 0, 64, M.PREC_F64)
    assert e.value.status == 8
    p = W.config2_problem(depth=3)
    p.n_substeps = B.MAX_SUBSTEPS + 1
    with pytest.raises(M.MpcError) as e:
        solver.eval_seeded(p, W.SEED, 0, 64, M.PREC_F64)
    assert e.value.status == 8


def test_everything_ties(solver):
    """all weights zero: every candidate scores exactly 0, far more than the 1024 near-ties the fast path lists -- the
    decision falls back to the strict kernel / strict expansion, and the reference's first-strict-minimum rule picks
    the FIRST candidate (controller.cpp:158-166)."""
    p = W.config2_problem(depth=4)
    p.w_distance = p.w_heading = p.w_speed = 0.0
    for prec in (M.PREC_FAST, M.PREC_F64):
        res = solver.eval_seeded(p, W.SEED, 7, 5000, prec)
        assert res.best_ordinal == 7 and res.best_last == 0.0
    # the tree: level l's winner is leaf 0 at every level, whichever model expands it
    q, _plan = W.config1_problem(depth=4)
    q.w_distance = q.w_heading = q.w_speed = 0.0
    for prec in (M.PREC_FAST, M.PREC_F64):
        res, levels = solver.solve_tree(q, 4, precision=prec)
        assert res.levels == 4 and res.best_ordinal == 0
        assert [levels[l].best_leaf for l in range(4)] == [0, 0, 0, 0]
        assert (res.first_rudder, res.first_throttle) == (q.min_rudder, q.min_throttle)     # first child: (minR, minT)
