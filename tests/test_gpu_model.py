"""mpc_hip_set_model: the 11 boat parameters of VehicleState.h:86-96 as per-handle / per-problem data.

* The default hull reproduces, bit for bit, what the library returned when those parameters were compile-time
  constants: tests/golden/gpu_bits_default_hull.json holds SHA-256 digests written by tools/gpu_bits.py from a build
  of the last commit before this entry point existed (candidate sets in both precisions and all schedule flavours,
  explicit controls, a fleet, both tree expansions, single rollouts, fleet current estimation).
* Another hull is checked against the oracle, which takes the same 11 parameters (orc_set_model): every candidate's
  cost, the argmin, the tree search level by level, a fleet that mixes hulls, current estimation.
"""
import importlib.util
import json
import math
import os

import numpy as np
import pytest

import mpc_amd as M
from mpc_amd import workloads as W
import oracle as O
from helpers import oracle_args, oracle_seeded, rel_err, traj_array

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TOL_FAST, TOL_F64 = 1e-5, 1e-12

# a heavier, faster vessel with a deeper rudder and a non-zero idle speed of the shaft
HULL = dict(idle_rpm=300.0, max_rpm=2800.0, max_rpm_change_rate=800.0, prop_ratio=0.45, prop_pitch=18.0,
            max_rudder_angle=0.6, rudder_coefficient=0.2, rudder_distance=2.5, mass=3500.0, max_power=12000.0, max_speed=3.5)


def _torch():
    import torch
    assert torch.cuda.is_available()
    return torch


def _gpu_bits():
    spec = importlib.util.spec_from_file_location("gpu_bits", os.path.join(ROOT, "tools", "gpu_bits.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


@pytest.fixture
def oracle_hull():
    O.set_model(**HULL)
    yield
    O.set_model()


@pytest.mark.parametrize("explicit_default", [False, True])
def test_default_hull_gives_the_bits_of_the_compile_time_constants(explicit_default):
    """The strict and tree digests are those of the build with compile-time model constants (commit 74424a6); the fast
    candidate-set digests were re-recorded when the fast model's spelling changed (the fixture's "provenance")."""
    want = json.load(open(os.path.join(ROOT, "tests", "golden", "gpu_bits_default_hull.json")))["digests"]
    setup = (lambda s: s.set_model(M.default_model())) if explicit_default else None
    got = _gpu_bits().digests(M, W, setup)
    assert sorted(got) == sorted(want) and len(want) >= 19
    for k in want:
        assert got[k] == want[k], k


def test_another_hull_candidate_sets_match_the_oracle(oracle_hull):
    torch = _torch()
    solver = M.Solver(0)
    try:
        solver.set_model(M.default_model(**HULL))
        for p, n in ((W.config3_problem(depth=12), 8192),
                     (M.make_problem((0.5, -0.5, 1.5, 1.8), W.config2_plan().checkpoints(W.T0, 8), M.substep_schedule(1.0),
                                     last=(0.1, 0.6), current=(0.2, -0.1)), 4096)):
            o_last, o_cum, o_best = oracle_seeded(p, W.SEED, 0, n)
            for prec, tol in ((M.PREC_F64, TOL_F64), (M.PREC_FAST, TOL_FAST)):
                last = torch.empty(n, dtype=torch.float64, device="cuda")
                cum = torch.empty(n, dtype=torch.float64, device="cuda")
                res = solver.eval_seeded(p, W.SEED, 0, n, prec, M.SELECT_LAST_STEP, last, cum)
                e1, e2 = rel_err(last.cpu().numpy(), o_last).max(), rel_err(cum.cpu().numpy(), o_cum).max()
                print("hull, precision %d: max rel err last %.3e cum %.3e" % (prec, e1, e2))
                assert e1 <= tol and e2 <= tol
                assert res.best_ordinal == o_best.best_index
        # ... and differ from the reference hull's (the setter is not a no-op)
        O.set_model()
        d_last, _, _ = oracle_seeded(p, W.SEED, 0, 64)
        assert rel_err(d_last, o_last[:64]).max() > 1e-3
    finally:
        solver.close()


def test_another_hull_tree_search_matches_the_oracle(oracle_hull):
    solver = M.Solver(0)
    try:
        solver.set_model(M.default_model(**HULL))
        w = W.DubinsWrapper((0.0, 0.0, math.pi / 2), (0.0, math.pi, 0.0), 8.0, W.LRL, 2.5, 4.0)
        w.end_time = 4.0 + 5.5
        plan = W.DubinsPlan([w])
        p = M.make_problem((0.0, 0.0, 0.0, 2.5), plan.checkpoints(4.0, 4), M.substep_schedule(1.0), last=(0.0, 0.0), current=(0.2, -0.1))
        cfg, veh, cps, sub = oracle_args(p)
        it, traj, olev, nodes = O.mpc_tree(cfg, O.State(p.start_x, p.start_y, p.start_heading, p.start_speed, 0.0), cps, 4, -1)
        results = []
        for prec in (M.PREC_F64, M.PREC_FAST):
            res, levels = solver.solve_tree(p, 4, precision=prec)
            assert res.levels == it == 4
            for l in range(4):
                assert (levels[l].n_leaves, levels[l].best_leaf) == (olev[l].n_leaves, olev[l].best_leaf), (prec, l)
                assert rel_err(levels[l].best_score, olev[l].best_score) <= TOL_F64
            got = traj_array(res)
            for l in range(5):
                want = [traj[l].v.s.x, traj[l].v.s.y, traj[l].v.s.heading, traj[l].v.s.speed, traj[l].v.rpm]
                assert rel_err(got[l, :5], want).max() <= TOL_F64, l
            assert (res.first_rudder, res.first_throttle) == (traj[1].last_rudder, traj[1].last_throttle)
            results.append(traj_array(res).tobytes())
        assert results[0] == results[1]                  # both expansions return the strict path's bits
    finally:
        solver.close()


def test_a_fleet_that_mixes_hulls():
    """models[p % 2]: even vessels keep the reference's hull, odd ones get HULL; each equals the oracle under its own"""
    solver = M.Solver(0)
    try:
        solver.set_model([M.default_model(), M.default_model(**HULL)])
        probs, seeds = W.fleet_problems(8, depth=10)
        n = 2048
        res = solver.eval_seeded_batch(probs, seeds, n, M.PREC_FAST)
        for k in range(8):
            O.set_model(**(HULL if k % 2 else {}))
            try:
                _, _, o_best = oracle_seeded(probs[k], seeds[k], 0, n)
            finally:
                O.set_model()
            assert res[k].status == 0 and res[k].best_ordinal == o_best.best_index, k
            assert rel_err(res[k].best_last, o_best.best_last) <= TOL_FAST
        # fleet current estimation replays with each vessel's own hull
        hist, actual = [], []
        for k in range(4):
            h = [(1.0 * k + 0.2 * i, -2.0 * k + 0.1 * i, (0.3 * k + 0.01 * i) % (2 * math.pi), 1.0 + 0.02 * i, 50.0 + 0.1 * i,
                  0.1 * k - 0.2, 0.5 + 0.1 * k) for i in range(30)]
            hist.append(h)
            actual.append((h[-1][0] + 0.5, h[-1][1] - 0.25, h[-1][4] + 0.1))
        got = solver.estimate_currents(hist, actual)
        for k in range(4):
            O.set_model(**(HULL if k % 2 else {}))
            try:
                entries = [O.HistoryEntry(O.State(*e[:5]), e[5], e[6]) for e in hist[k]]
                want = O.current_estimate(entries, True, O.State(actual[k][0], actual[k][1], 0.0, 0.0, actual[k][2]))
            finally:
                O.set_model()
            assert rel_err(np.array(got[k]), np.array(want)).max() <= TOL_F64, k
    finally:
        solver.close()


def test_hulls_outside_the_fast_envelope_and_invalid_hulls():
    solver = M.Solver(0)
    try:
        p = W.config2_problem(depth=4)
        solver.set_model(M.default_model(max_rudder_angle=1.2))          # 69 degrees: beyond the fast model's polynomials
        with pytest.raises(M.MpcError) as e:
            solver.eval_seeded(p, W.SEED, 0, 256, M.PREC_FAST)
        assert e.value.status == 1 and "MPC_PREC_F64" in str(e.value)
        with pytest.raises(M.MpcError):
            solver.solve_tree(p, 2, precision=M.PREC_FAST)
        O.set_model(max_rudder_angle=1.2)
        try:
            o_last, _, o_best = oracle_seeded(p, W.SEED, 0, 256)
        finally:
            O.set_model()
        torch = _torch()
        last = torch.empty(256, dtype=torch.float64, device="cuda")
        res = solver.eval_seeded(p, W.SEED, 0, 256, M.PREC_F64, M.SELECT_LAST_STEP, last, None)       # the strict model takes any hull
        assert rel_err(last.cpu().numpy(), o_last).max() <= TOL_F64 and res.best_ordinal == o_best.best_index
        for bad in (dict(mass=0.0), dict(max_rpm=0.0), dict(prop_pitch=-1.0), dict(max_speed=float("nan")),
                    dict(max_speed=62.2568 ** 0.5 * 0 + (3200 * 0.389105058 / 20))):       # max speed == max prop speed: :100 divides by 0
            with pytest.raises(M.MpcError) as e:
                solver.set_model(M.default_model(**bad))
            assert e.value.status == 1, bad
        # a new hull drops the resident problems
        solver.set_model(M.default_model())
        solver.upload([p], [W.SEED])
        solver.set_model(M.default_model(**HULL))
        with pytest.raises(M.MpcError):
            solver.run_seeded(0, 256)
    finally:
        solver.close()
