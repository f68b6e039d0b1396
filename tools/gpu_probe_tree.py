"""perf probe (not a test, not the bench): the reference-native tree search at growing depth, GPU vs CPU oracle."""
import math
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import time
import torch  # noqa: F401
import mpc_amd as M
from mpc_amd import workloads as W
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from helpers import oracle_args
import oracle as O

s = M.Solver(0)
w = W.DubinsWrapper((0.0, 0.0, math.pi / 2), (0.0, math.pi, 0.0), 8.0, W.LRL, 2.5, 4.0)
plan = W.DubinsPlan([w])
PREC = M.PREC_FAST if len(sys.argv) > 1 and sys.argv[1] == "fast" else M.PREC_F64
for depth in (3, 4, 5, 6, 7):
    p = M.make_problem((0.0, 0.0, 0.0, 2.5), plan.checkpoints(4.0, depth), M.substep_schedule(1.0), current=(0.2, -0.1))
    for rep in range(3):
        t0 = time.perf_counter()
        try:
            res, lv = s.solve_tree(p, depth, precision=PREC, max_nodes=1 << 30)
        except M.MpcError as e:
            print("depth", depth, "failed:", e); break
        wall = (time.perf_counter() - t0) * 1e3
    line = ("fast " if PREC == M.PREC_FAST else "") + "depth=%d levels=%d leaves=%s evaluated=%d gpu_ms=%.3f wall_ms=%.3f steps/s=%.3e first=(%g,%g)" % (
        depth, res.levels, [int(lv[i].n_leaves) for i in range(res.levels)][-1], res.evaluated, res.total_ms, wall,
        res.rollout_steps / (res.total_ms * 1e-3), res.first_rudder, res.first_throttle)
    if depth <= 5:
        cfg, veh, cps, sub = oracle_args(p)
        t0 = time.perf_counter()
        it, traj, olev, nodes = O.mpc_tree(cfg, O.State(0, 0, 0, 2.5, 0), cps, depth)
        cpu = (time.perf_counter() - t0) * 1e3
        line += " | cpu_ms=%.1f cpu_nodes=%d same=%s" % (cpu, nodes, (traj[1].last_rudder, traj[1].last_throttle) == (res.first_rudder, res.first_throttle))
    print(line, flush=True)
