"""Product-side input construction (mpc_amd/workloads.py) against the oracle's independent C restatement and the
committed known answers.  CPU only."""
import json
import math
import os

import numpy as np

import mpc_amd as M
from mpc_amd import workloads as W
import oracle as O

GOLD = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "candidates.json")))


def _oracle_plan(plan):
    ws = []
    for w in plan.wrappers:
        ws.append(O.dubins(w.qi, w.param, w.rho, w.kind, w.speed, w.start_time, end_time=w.end_time))
    return O.plan_array(ws)


def test_dubins_sampling_equals_the_oracle_bit_for_bit():
    for plan in (W.config2_plan(), W.config3_plan()):
        op = _oracle_plan(plan)
        for t in np.linspace(plan.start_time() - 1.0, plan.end_time() + 1.0, 257):
            s = O.plan_sample(op, float(t))
            assert plan.sample(float(t)) == (s.x, s.y, s.heading, s.speed), t
        assert plan.total_time() == O.lib().orc_plan_total_time(op, len(op))
        assert plan.max_iterations(W.T0) == O.lib().orc_max_iterations(op, len(op), W.T0)


def test_all_six_dubins_words():
    for kind in range(6):
        mid = 4.0 if kind in (W.RLR, W.LRL) else 2.0
        w = W.DubinsWrapper((3.0, -2.0, 0.7), (0.9, mid, 0.6), 8.0, kind, 2.5, 10.0)
        ow = O.dubins(w.qi, w.param, w.rho, kind, w.speed, w.start_time)
        assert w.end_time == ow.end_time
        for a in np.linspace(0, sum(w.param) * w.rho, 33):
            q = (O.C.c_double * 3)()
            O.lib().orc_dubins_sample(O.C.byref(ow), float(a), q)
            assert w.pose_at_length(float(a)) == (q[0], q[1], q[2])


def test_plans_are_continuous_and_long_enough():
    for plan, need in ((W.config2_plan(), 31.0), (W.config3_plan(), 21.0)):
        assert plan.total_time() >= need
        prev = plan.sample(W.T0)
        for t in np.arange(W.T0 + 0.05, W.T0 + need, 0.05):
            cur = plan.sample(float(t))
            assert math.hypot(cur[0] - prev[0], cur[1] - prev[1]) < 0.05 * 2.5 * 1.01     # no jumps between words
            prev = cur


def test_seeded_controls_known_answers():
    for c in GOLD["cases"]:
        r, t = W.seeded_controls(c["seed"], c["ordinal"], 1, c["level"] + 1, *c["last"])
        assert (r[c["level"], 0], t[c["level"], 0]) == (c["rudder"], c["throttle"]), c
        assert O.lib().orc_seeded_hash(c["seed"], c["ordinal"], c["level"]) == c["hash"]


def test_seeded_controls_numpy_equals_c_header():
    seed, first, count, depth = W.SEED, 65530, 40, 30
    rud, thr = W.seeded_controls(seed, first, count, depth, 0.125, 0.5)
    r, t = O.C.c_double(), O.C.c_double()
    for i in range(count):
        for l in range(depth):
            O.lib().orc_seeded_control(seed, first + i, l, 0.125, 0.5, O.C.byref(r), O.C.byref(t))
            assert (rud[l, i], thr[l, i]) == (r.value, t.value)
    assert np.all(np.abs(rud) <= 1) and np.all((thr >= 0) & (thr <= 1))
    assert np.array_equal(rud.astype(np.float32).astype(np.float64), rud)          # exactly representable in FP32


def test_config_problems_are_wellformed():
    p2, p3 = W.config2_problem(), W.config3_problem()
    assert (p2.depth, p2.n_substeps) == (20, 20) and (p3.current_x, p3.current_y) == (0.35, -0.20)
    assert abs(sum(p2.substeps[i] for i in range(20)) - 1.0) < 1e-12
    assert W.config2_problem(30).depth == 30
    p1, plan = W.config1_problem(depth=5)
    assert p1.n_substeps == 11 and p1.depth == 5
    probs, seeds = W.fleet_problems(16, depth=20)
    assert len(probs) == 16 and len(set(seeds)) == 16 and all(q.depth == 20 for q in probs)
    # a fleet checkpoint list must really lie on its plan: consecutive checkpoints 2.5 m apart (1 s at 2.5 m/s) or less
    for q in probs:
        for i in range(1, q.depth):
            d = math.hypot(q.checkpoints[i].x - q.checkpoints[i - 1].x, q.checkpoints[i].y - q.checkpoints[i - 1].y)
            assert d <= 2.5 * 1.0001
