"""Device-side reference-trajectory sampling (mpc_hip_upload_problems_from_plans) against the oracle's
DubinsPlan::sample restatement, and a fleet solved from plans against the same fleet solved from host checkpoints."""
import numpy as np
import pytest

import mpc_amd as M
from mpc_amd import workloads as W
from helpers import oracle_seeded, rel_err
import oracle as O

pytestmark = pytest.mark.gpu


def _oracle_plan(plan):
    return O.plan_array([O.dubins(w.qi, w.param, w.rho, w.kind, w.speed, w.start_time) for w in plan.wrappers])


def test_all_six_words_clamping_and_gaps(solver):
    """every Dubins word, times before the plan, after it and inside a gap between two wrappers"""
    depth = 12
    plans, t0s = [], []
    for kind in range(6):
        mid = 4.0 if kind in (W.RLR, W.LRL) else 2.0
        first = W.DubinsWrapper((3.0 * kind, -7.0, 0.4 + kind), (0.7, mid, 0.5), 8.0, kind, 2.5, 100.0)
        # second wrapper starts 1.5 s after the first ends: a gap
        second = W.DubinsWrapper(first.end_pose(), (0.0, 1.5, 0.0), 8.0, W.LSL, 2.0, first.end_time + 1.5)
        plans.append(W.DubinsPlan([first, second]))
        t0s.append(100.0 - 2.5 + kind)             # the first checkpoints lie before the plan starts
    probs = [W.config2_problem(depth) for _ in plans]
    for p in probs:
        p.current_x, p.current_y = 0.3, -0.2
    solver.upload_from_plans(probs, [W.SEED] * len(plans), [pl.to_struct() for pl in plans], t0s)
    for k, (plan, t0) in enumerate(zip(plans, t0s)):
        op = _oracle_plan(plan)
        got = np.array(solver.fetch_checkpoints(k, depth))
        want = np.array([(s.x, s.y, s.heading, s.speed) for s in (O.plan_sample(op, t0 + i + 1.0) for i in range(depth))])
        # ocml vs libm sin/cos: a few ulps of a 100-m coordinate; heading compared on the circle
        assert np.abs(got[:, :2] - want[:, :2]).max() <= 1e-12, k
        dh = np.abs(got[:, 2] - want[:, 2]); dh = np.minimum(dh, 2 * np.pi - dh)
        assert dh.max() <= 1e-13 and (got[:, 3] == want[:, 3]).all(), k
        assert (got[:, 2] >= 0).all() and (got[:, 2] < 2 * np.pi).all()


def test_fleet_from_plans_equals_fleet_from_host_checkpoints(solver):
    n_problems, n, depth = 96, 2048, 20
    probs, seeds, plans = W.fleet_problems(n_problems, depth, with_plans=True)
    ref = solver.eval_seeded_batch(probs, seeds, n, M.PREC_F64)
    ref = [(r.best_ordinal, r.best_last) for r in ref]
    blank = []
    for p in probs:                                   # the device must not depend on the host checkpoints
        q = M.Problem.from_buffer_copy(p)
        for i in range(depth):
            q.checkpoints[i].x = q.checkpoints[i].y = q.checkpoints[i].heading = q.checkpoints[i].speed = float("nan")
        blank.append(q)
    solver.upload_from_plans(blank, seeds, [pl.to_struct() for pl in plans], [W.T0] * n_problems)
    for prec, tol in ((M.PREC_F64, 1e-11), (M.PREC_FAST, 1e-5)):
        solver.run_seeded(0, n, prec)
        res = solver.fetch()
        assert all(r.status == 0 for r in res)
        for k, r in enumerate(res):
            assert r.best_ordinal == ref[k][0], (prec, k)
            assert rel_err(r.best_last, ref[k][1]) <= tol, (prec, k)
    # one vessel completely against the oracle, from the device's own checkpoints
    k = 17
    cps = solver.fetch_checkpoints(k, depth)
    p = M.Problem.from_buffer_copy(probs[k])
    for i, c in enumerate(cps):
        p.checkpoints[i] = M.Checkpoint(*c)
    o_last, _, o_best = oracle_seeded(p, seeds[k], 0, n)
    solver.run_seeded(0, n, M.PREC_F64)
    r = solver.fetch()[k]
    assert r.best_ordinal == o_best.best_index and rel_err(r.best_last, o_best.best_last) <= 1e-12


def test_bad_plans_are_refused(solver):
    p = W.config2_problem(4)
    empty = M.Plan()
    with pytest.raises(M.MpcError) as e:
        solver.upload_from_plans([p], [1], [empty], [0.0])
    assert e.value.status == 4 and "empty plan" in str(e.value)          # controller.cpp:37
    bad = W.config2_plan().to_struct()
    bad.segments[0].type = 9
    with pytest.raises(M.MpcError):
        solver.upload_from_plans([p], [1], [bad], [0.0])
