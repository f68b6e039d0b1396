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


@pytest.mark.parametrize("depth", [28, 29, 32, 57])
def test_a_million_explicit_candidates_at_the_lds_bound(solver, depth):
    """>= 2^20 explicit candidates run 256-lane workgroups, whose staged controls (2 * depth * 256 floats) plus the
    7.3 KB of problem data pass 64 KiB of LDS from depth 29 on: the library then halves the workgroup (128 lanes up to
    depth 56, 64 beyond).  4096 distinct control sequences tiled 256 times, all checked against the oracle; the first
    occurrence of the best sequence wins."""
    torch = _torch()
    p = _circle_problem(depth, W.uniform_substeps())
    base, reps = 4096, 256
    n = base * reps
    rud, thr = W.seeded_controls(W.SEED + depth, 0, base, depth, p.last_rudder, p.last_throttle)
    rud32, thr32 = rud.astype(np.float32), thr.astype(np.float32)
    d_rud = torch.tensor(rud32, device="cuda").repeat(1, reps).contiguous()           # [depth][n]: lane i = sequence i % 4096
    d_thr = torch.tensor(thr32, device="cuda").repeat(1, reps).contiguous()
    o_last, o_cum, o_best = oracle_explicit(p, rud32.astype(np.float64), thr32.astype(np.float64))
    last = torch.empty(n, dtype=torch.float64, device="cuda")
    cum = torch.empty(n, dtype=torch.float64, device="cuda")
    res = solver.eval_candidates(p, d_rud, d_thr, n, M.PREC_FAST, M.SELECT_LAST_STEP, last, cum)
    got = last.cpu().numpy().reshape(reps, base)
    assert (got == got[0]).all()                                 # a sequence's cost does not depend on where it sits
    assert rel_err(got[0], o_last).max() <= TOL_FAST
    assert rel_err(cum.cpu().numpy()[:base], o_cum).max() <= TOL_FAST
    assert res.status == 0 and res.best_ordinal == o_best.best_index       # lowest ordinal among 256 exact ties
    assert res.rollout_steps == n * depth * 20
    # strict kernel, same launch shape
    res64 = solver.eval_candidates(p, d_rud, d_thr, n, M.PREC_F64, M.SELECT_LAST_STEP, last, None)
    assert rel_err(last.cpu().numpy()[:base], o_last).max() <= TOL_F64 and res64.best_ordinal == o_best.best_index


def test_negative_or_non_finite_weights_are_rejected(solver):
    """scores are ordered through their bit patterns, which needs them >= 0: weights outside [0, inf) are an argument
    error (cfg/mpc.cfg:11-15 allows [0, 10], [0, 60], [0, 10]), not a silently arbitrary winner"""
    for bad in ((-1.0, 20.0, 5.0), (1.0, float("nan"), 5.0), (1.0, 20.0, float("inf"))):
        p = _circle_problem(4, W.uniform_substeps())
        p.w_distance, p.w_heading, p.w_speed = bad
        with pytest.raises(M.MpcError) as e:
            solver.eval_seeded(p, W.SEED, 0, 256)
        assert e.value.status == 1 and "weights" in str(e.value)
        with pytest.raises(M.MpcError):
            solver.solve_tree(p, 2)
    p = _circle_problem(4, W.uniform_substeps())
    p.w_distance, p.w_heading, p.w_speed = 0.0, 0.0, 0.0                    # zeros are fine: everything ties at 0, ordinal 0 wins
    assert solver.eval_seeded(p, W.SEED, 0, 256).best_ordinal == 0


def test_misuse_of_the_stream_and_multi_gpu_entry_points_is_refused():
    """pipeline level out of range, winner records before any run / shallower than a resident problem / of a depth out
    of range, exact combine with too many shards or a row count that is not a multiple of the resident problems: every
    one an MPC_ERR_* with a message, none a launch"""
    import torch
    s = M.Solver(0)
    try:
        with pytest.raises(M.MpcError):
            s.set_pipeline(5)
        p = W.config2_problem(depth=6)
        s.upload([p, p], [1, 2])
        buf = torch.zeros(4096, dtype=torch.int64, device="cuda")
        with pytest.raises(M.MpcError):
            s.copy_winner_records(buf, 6)                     # nothing has run
        s.run_seeded(0, 256, M.PREC_FAST)
        with pytest.raises(M.MpcError):
            s.copy_winner_records(buf, 5)                     # shallower than the problems
        with pytest.raises(M.MpcError):
            s.copy_winner_records(buf, M.MAX_DEPTH + 1)
        with pytest.raises(M.MpcError):
            s.set_winner_record_sink(buf, 0)
        out = torch.zeros(64, dtype=torch.int64, device="cuda")
        with pytest.raises(M.MpcError):
            s.combine_shards_exact(buf, 65, 2, 6, 0, out)     # a lane per shard: at most 64
        with pytest.raises(M.MpcError):
            s.combine_shards_exact(buf, 2, 3, 6, 0, out)      # rows must be solves x problems
        with pytest.raises(M.MpcError):
            s.combine_shards_exact(buf, 2, 2, 5, 0, out)      # records shallower than the problems
        with pytest.raises(M.MpcError):
            s.combine_shards_exact(buf, 2, 2, 6, 0, out, precision=7)        # the records' precision / selection are arguments
        with pytest.raises(M.MpcError):
            s.combine_shards_exact(buf, 2, 2, 6, 0, out, selection=-1)
        # and the well-formed call right after still works: two shards holding the same records -> the lower shard wins ties
        s.copy_winner_records(buf[:2 * M.winner_record_words(6)], 6)
        W6 = M.winner_record_words(6)
        buf[2 * W6:4 * W6] = buf[:2 * W6]
        torch.cuda.synchronize()
        s.combine_shards_exact(buf, 2, 2, 6, 1000, out)
        s.synchronize()
        res = s.fetch()
        o = out[:4].view(2, 2).cpu().numpy()
        assert [int(o[q, 1]) for q in range(2)] == [r.best_ordinal for r in res]
        assert s.tied_rows() == 2                             # both rows: two shards at the same fast score
    finally:
        s.close()
