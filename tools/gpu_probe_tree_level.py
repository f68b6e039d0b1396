"""profiling probe (not a test, not the bench): ONE depth of the reference-native tree search, a few repetitions, so that
a rocprofv3 --pmc pass over it stays short:  python tools/gpu_probe_tree_level.py fast|strict [depth=7] [reps=2]"""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: F401
import mpc_amd as M
from mpc_amd import workloads as W

prec = M.PREC_FAST if len(sys.argv) > 1 and sys.argv[1] == "fast" else M.PREC_F64
depth = int(sys.argv[2]) if len(sys.argv) > 2 else 7
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 2
s = M.Solver(0)
p, d = W.semicircle_problem(depth + 1.5)
assert d == depth
for rep in range(reps):
    t0 = time.perf_counter()
    res, lv = s.solve_tree(p, depth, precision=prec)
    wall = (time.perf_counter() - t0) * 1e3
    print("%s depth=%d leaves=%d evaluated=%d gpu_ms=%.3f wall_ms=%.3f steps/s=%.3e first=(%g,%g) tie=%d" % (
        "fast" if prec == M.PREC_FAST else "strict", depth, lv[depth - 1].n_leaves, res.evaluated, res.total_ms, wall,
        res.rollout_steps / (res.total_ms * 1e-3), res.first_rudder, res.first_throttle, res.libm_tie), flush=True)
