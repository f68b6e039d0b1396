"""mpc_hip_solve_tree_deep on one GPU (python tools/gpu_probe_deep.py [fast|strict] [depth] [reps] [scenario]): wall per call."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mpc_amd as M
from mpc_amd import workloads as W

mode = sys.argv[1] if len(sys.argv) > 1 else "fast"
depth = int(sys.argv[2]) if len(sys.argv) > 2 else 8
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
scenario = sys.argv[4] if len(sys.argv) > 4 else "semicircle"
prec = M.PREC_FAST if mode == "fast" else M.PREC_F64
p = W.config1_problem(depth=depth)[0] if scenario == "config1" else W.semicircle_problem(depth + 1.5)[0]
solver = M.Solver(0)
for rep in range(reps):
    t0 = time.perf_counter()
    res, lv = solver.solve_tree_deep(p, depth, precision=prec)
    dt = (time.perf_counter() - t0) * 1e3
    print("%s %s deep depth=%d levels=%d leaves=%d wall_ms=%.2f first=(%g,%g) ordinal=%d tie=%d"
          % (scenario, mode, depth, res.levels, lv[res.levels - 1].n_leaves, dt, res.first_rudder, res.first_throttle, res.best_ordinal, res.libm_tie), flush=True)
