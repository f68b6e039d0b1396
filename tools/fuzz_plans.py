"""A long random run of the device-side DubinsPlan::sample against the oracle's sampler (python tools/fuzz_plans.py [plans] [seed]):
plans of 1-4 random Dubins words (all six kinds, random radii, speeds, gaps between wrappers), sampled at 24 checkpoint
times that start before the plan and run past its end."""
import math
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import mpc_amd as M
from mpc_amd import workloads as W
import oracle as O

n_plans = int(sys.argv[1]) if len(sys.argv) > 1 else 500
rng = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 41)
solver = M.Solver(0)
depth = 24
plans, t0s = [], []
for k in range(n_plans):
    t = float(rng.uniform(0.0, 1000.0))
    pose = (float(rng.uniform(-500, 500)), float(rng.uniform(-500, 500)), float(rng.uniform(0, 2 * math.pi)))
    wrappers = []
    for j in range(int(rng.randint(1, 5))):
        kind = int(rng.randint(0, 6))
        mid = float(rng.uniform(math.pi + 0.1, 2 * math.pi - 0.1)) if kind in (W.RLR, W.LRL) else float(rng.uniform(0.0, 6.0))
        w = W.DubinsWrapper(pose, (float(rng.uniform(0.0, 3.0)), mid, float(rng.uniform(0.0, 3.0))), float(rng.uniform(2.0, 30.0)), kind,
                            float(rng.uniform(0.5, 3.0)), t)
        wrappers.append(w)
        pose = w.end_pose()
        t = w.end_time + float(rng.choice([0.0, 0.0, rng.uniform(0.1, 3.0)]))
    plans.append(W.DubinsPlan(wrappers))
    t0s.append(wrappers[0].start_time - float(rng.uniform(0.0, 4.0)))
worst_xy, worst_h, speed_bits = 0.0, 0.0, True
for base in range(0, n_plans, 256):
    chunk = plans[base:base + 256]
    probs = [W.config2_problem(depth) for _ in chunk]
    solver.upload_from_plans(probs, [W.SEED] * len(chunk), [pl.to_struct() for pl in chunk], t0s[base:base + 256])
    for k, plan in enumerate(chunk):
        op = O.plan_array([O.dubins(w.qi, w.param, w.rho, w.kind, w.speed, w.start_time) for w in plan.wrappers])
        got = np.array(solver.fetch_checkpoints(k, depth))
        want = np.array([(s.x, s.y, s.heading, s.speed) for s in (O.plan_sample(op, t0s[base + k] + i + 1.0) for i in range(depth))])
        worst_xy = max(worst_xy, float(np.abs(got[:, :2] - want[:, :2]).max()))
        dh = np.abs(got[:, 2] - want[:, 2]); dh = np.minimum(dh, 2 * np.pi - dh)
        worst_h = max(worst_h, float(dh.max()))
        speed_bits &= bool((got[:, 3] == want[:, 3]).all())
print("plans", n_plans, "x", depth, "checkpoints; worst position difference %.3g m, worst heading difference %.3g rad, speeds bit-identical: %s"
      % (worst_xy, worst_h, speed_bits))
