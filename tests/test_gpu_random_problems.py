"""Randomly drawn problems (fixed seed): weights, currents, start states, references, substep schedules and depths
that none of the BASELINE configurations use, every candidate's cost against the oracle and the same chosen control.
The schedules cover each flavour the kernels specialise for: all substeps equal, equal but for the last (what
VehicleState::simulate itself produces), and ragged."""
import math

import numpy as np
import pytest

import mpc_amd as M
from mpc_amd import binding as B
from mpc_amd import workloads as W
from helpers import oracle_seeded, rel_err

pytestmark = pytest.mark.gpu

TOL = {M.PREC_F64: 1e-12, M.PREC_FAST: 1e-5}


def _schedule(rng):
    kind = rng.randint(0, 6)
    if kind == 0:
        return W.uniform_substeps()                              # 20 x 0.05 s
    if kind == 1:
        return [0.1] * 10
    if kind == 2:
        return list(B.substep_schedule(1.0))                     # 10 x 0.1 s + 1.4e-16 s: the reference's own level
    if kind == 3:
        return list(B.substep_schedule(float(rng.uniform(0.15, 2.5))))        # k x 0.1 s + a remainder
    if kind == 4:
        return [1.0 / 64] * 64
    return [float(v) for v in rng.uniform(0.005, 0.1, size=rng.randint(1, 30))]   # ragged


def _problem(rng):
    depth = int(rng.randint(1, 25))
    sub = _schedule(rng)
    level_time = sum(sub)
    speed = float(rng.uniform(0.5, 2.75))
    words = [(int(rng.randint(0, 4)), (float(rng.uniform(0.1, 1.5)), float(rng.uniform(0.5, 4.0)), float(rng.uniform(0.1, 1.5)))),
             (W.LSL, (0.0, 40.0, 0.0))]
    plan = W.chain(words, float(rng.uniform(6.0, 20.0)), speed, (float(rng.uniform(-100, 100)), float(rng.uniform(-100, 100)),
                                                                  float(rng.uniform(0, 2 * math.pi))), W.T0)
    cps = [plan.sample(W.T0 + (i + 1) * level_time) for i in range(depth)]
    x, y, heading, _ = plan.sample(W.T0)
    start = (x + float(rng.uniform(-3, 3)), y + float(rng.uniform(-3, 3)), (heading + float(rng.uniform(-0.6, 0.6))) % (2 * math.pi),
             float(rng.choice([0.0, rng.uniform(0.0, 2.75), rng.uniform(2.75, 3.5)])))       # at rest, cruising, faster than the model's top speed
    p = B.make_problem(start, cps, sub, last=(float(rng.randint(-16, 17)) * 0.0625, float(rng.randint(0, 9)) * 0.125),
                       current=(float(rng.uniform(-0.8, 0.8)), float(rng.uniform(-0.8, 0.8))))
    w = rng.uniform(0.0, 30.0, size=3) * (rng.uniform(size=3) > 0.15)                        # some weights exactly zero
    if w[0] == 0.0 and w[1] == 0.0 and w[2] == 0.0:
        w[0] = 1.0
    p.w_distance, p.w_heading, p.w_speed = float(w[0]), float(w[1]), float(w[2])
    return p


@pytest.mark.parametrize("block", range(4))
def test_random_problems_every_cost_and_the_choice(solver, block):
    torch = pytest.importorskip("torch")
    rng = np.random.RandomState(1000 + block)
    n = 1024
    for k in range(10):
        p = _problem(rng)
        seed = int(rng.randint(1, 2 ** 31))
        first = int(rng.randint(0, 5000))
        o_last, o_cum, o_best = oracle_seeded(p, seed, first, n)
        for prec in (M.PREC_F64, M.PREC_FAST):
            if prec == M.PREC_FAST and max(p.substeps[i] for i in range(p.n_substeps)) > 0.1000001:
                continue
            last = torch.empty(n, dtype=torch.float64, device="cuda")
            cum = torch.empty(n, dtype=torch.float64, device="cuda")
            res = solver.eval_seeded(p, seed, first, n, prec, M.SELECT_LAST_STEP, last, cum)
            what = (block, k, prec, p.depth, p.n_substeps)
            assert rel_err(last.cpu().numpy(), o_last).max() <= TOL[prec], what
            assert rel_err(cum.cpu().numpy(), o_cum).max() <= TOL[prec], what
            assert res.best_ordinal == o_best.best_index, what
            assert res.levels == p.depth
