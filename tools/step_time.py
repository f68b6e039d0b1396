"""Wall-clock per solve of config 2 (back-to-back run_candidates on one stream), independent of the library's events.
Usage: python tools/step_time.py [n_solves]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import mpc_amd as M
from mpc_amd import workloads as W

n = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
if os.environ.get("MPC_LIB"):                      # an experimental build of the library (exp/)
    M.binding._lib = M.binding.load_library(os.environ["MPC_LIB"])
depth, count = 20, 65536
stream = torch.cuda.Stream()
with torch.cuda.stream(stream):
    solver = M.Solver(0, stream=stream.cuda_stream)
    prob = W.config2_problem(depth)
    r, t = W.seeded_controls(W.SEED, 0, count, depth, prob.last_rudder, prob.last_throttle)
    d_rud = torch.tensor(r, dtype=torch.float32, device="cuda").contiguous()
    d_thr = torch.tensor(t, dtype=torch.float32, device="cuda").contiguous()
    solver.upload([prob], [W.SEED])
    for _ in range(100):
        solver.run_candidates(d_rud, d_thr, count, M.PREC_FAST, M.SELECT_LAST_STEP)
    torch.cuda.synchronize()
    solver.timing(reset=True)
    t0 = time.perf_counter()
    for _ in range(n):
        solver.run_candidates(d_rud, d_thr, count, M.PREC_FAST, M.SELECT_LAST_STEP)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) * 1e3 / n
    runs, k, tot = solver.timing(reset=True)
    res = solver.fetch()[0]
print("lib=%s  wall %.4f ms/solve  k1 %.4f ms  span %.4f ms  ordinal %d score %.12g" % (
    os.environ.get("MPC_LIB", "default"), ms, k / max(runs, 1), tot / max(runs, 1), res.best_ordinal, res.best_last))
