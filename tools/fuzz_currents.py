"""A long random run of the device-side CurrentEstimator::getCurrent against the oracle (python tools/fuzz_currents.py [vessels] [seed]):
random histories of 0-600 samples, sample rates, jitters, start states and true currents, uploaded and device-resident."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import mpc_amd as M
import oracle as O
import test_gpu_currents as T

n_vessels = int(sys.argv[1]) if len(sys.argv) > 1 else 400
rng = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 21)
solver = M.Solver(0)
t0 = time.time()
cases = []
for k in range(n_vessels):
    n = int(rng.choice([0, 5, 9, 10, 11, int(rng.randint(12, 600)), int(rng.randint(12, 600)), 600]))
    cur = (rng.uniform(-0.8, 0.8), rng.uniform(-0.8, 0.8))
    start = (rng.uniform(-200, 200), rng.uniform(-200, 200), rng.uniform(0, 6.28), float(rng.choice([0.0, rng.uniform(0, 3.2)])), rng.uniform(0, 3000))
    hist, v = T._history(rng, n, cur, hz=float(rng.choice([2.0, 5.0, 10.0, 20.0])), jitter=float(rng.choice([0.0, 0.005, 0.02])), start=start)
    cases.append((hist, v.s))
got = solver.estimate_currents([T._as_samples(h) for h, _ in cases], [(a.x, a.y, a.time) for _, a in cases])
worst, bad = 0.0, []
for k, ((h, a), g) in enumerate(zip(cases, got)):
    want = O.current_estimate(h, True, a)
    e = max(abs(g[0] - want[0]) / max(1.0, abs(want[0])), abs(g[1] - want[1]) / max(1.0, abs(want[1])))
    worst = max(worst, e)
    if e > 1e-12:
        bad.append((k, len(h), g, want))
print("vessels", n_vessels, "; worst error of an estimate (relative to max(1, |estimate|))", worst, "; beyond 1e-12:", len(bad), "; %.0f s" % (time.time() - t0))
for b in bad[:10]:
    print("   ", b)
