"""A long random run of the tree search against the oracle's DFS (python tools/fuzz_tree.py [problems] [seed]): random
start states, last controls, currents, weights (some zero), granularities and Dubins references; depths 2-4 (the oracle's
depth 4 costs 33 ms); both expansions.  Prints every disagreement instead of stopping at the first."""
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import mpc_amd as M
from mpc_amd import workloads as W
from helpers import oracle_args, rel_err, traj_array
import oracle as O

n_problems = int(sys.argv[1]) if len(sys.argv) > 1 else 200
rng = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 7)
solver = M.Solver(0)
bad, worst = [], 0.0
t0 = time.time()
for k in range(n_problems):
    depth = int(rng.randint(2, 5))
    speed = float(rng.uniform(0.5, 2.75))
    words = [(int(rng.randint(0, 4)), (float(rng.uniform(0.1, 1.5)), float(rng.uniform(0.5, 4.0)), float(rng.uniform(0.1, 1.5)))),
             (W.LSL, (0.0, 40.0, 0.0))]
    plan = W.chain(words, float(rng.uniform(6.0, 20.0)), speed, (float(rng.uniform(-100, 100)), float(rng.uniform(-100, 100)),
                                                                  float(rng.uniform(0, 2 * math.pi))), W.T0)
    cps = [plan.sample(W.T0 + (i + 1) * 1.0) for i in range(depth)]
    x, y, heading, _ = plan.sample(W.T0)
    start = (x + float(rng.uniform(-3, 3)), y + float(rng.uniform(-3, 3)), (heading + float(rng.uniform(-0.6, 0.6))) % (2 * math.pi),
             float(rng.choice([0.0, rng.uniform(0.0, 2.75), rng.uniform(2.75, 3.5)])))
    w = rng.uniform(0.0, 30.0, size=3) * (rng.uniform(size=3) > 0.15)
    if not w.any():
        w[0] = 1.0
    p = M.make_problem(start, cps, M.substep_schedule(1.0), last=(float(rng.randint(-16, 17)) * 0.0625, float(rng.randint(0, 9)) * 0.125),
                       current=(float(rng.uniform(-0.8, 0.8)), float(rng.uniform(-0.8, 0.8))), weights=tuple(float(v) for v in w))
    p.rudder_granularity = float(rng.choice([0.0625, 0.125, 0.25, 0.03125]))
    p.throttle_granularity = float(rng.choice([0.125, 0.25, 0.0625]))
    cfg, veh, ocps, sub = oracle_args(p)
    it, traj, olev, nodes = O.mpc_tree(cfg, O.State(p.start_x, p.start_y, p.start_heading, p.start_speed, 0.0), ocps, depth, -1)
    results = []
    for prec in (M.PREC_F64, M.PREC_FAST):
        res, levels = solver.solve_tree(p, depth, precision=prec)
        why = []
        if res.status != 0 or res.levels != it:
            why.append("status/levels %d %d vs %d" % (res.status, res.levels, it))
        for l in range(min(it, res.levels)):
            if levels[l].n_leaves != olev[l].n_leaves: why.append("leaves at level %d: %d vs %d" % (l, levels[l].n_leaves, olev[l].n_leaves))
            if levels[l].best_leaf != olev[l].best_leaf: why.append("leaf at level %d: %d vs %d (scores %.17g %.17g)" % (l, levels[l].best_leaf, olev[l].best_leaf, levels[l].best_score, olev[l].best_score))
            worst = max(worst, float(rel_err(levels[l].best_score, olev[l].best_score)))
        got = traj_array(res)
        for l in range(min(it, res.levels) + 1):
            want = [traj[l].v.s.x, traj[l].v.s.y, traj[l].v.s.heading, traj[l].v.s.speed, traj[l].v.rpm]
            e = float(rel_err(got[l, :5], want).max())
            if e > 1e-12: why.append("trajectory level %d off by %.3g" % (l, e))
            if got[l, 7] != traj[l].last_rudder or got[l, 8] != traj[l].last_throttle: why.append("controls at level %d" % l)
        if why:
            bad.append((k, prec, depth, tuple(float(v) for v in w), why[:3]))
        results.append((res.best_ordinal, res.best_last, res.best_cumulative, traj_array(res).tobytes()))
    if results[0] != results[1]:
        bad.append((k, "fast expansion differs from the strict one", depth))
    if k % 50 == 49:
        print("%d problems, %d disagreements, worst level-score error %.3g, %.0f s" % (k + 1, len(bad), worst, time.time() - t0), flush=True)
print("tree problems", n_problems, "disagreements", len(bad), "worst relative error of a level's best score", worst)
for b in bad:
    print("   ", b)
