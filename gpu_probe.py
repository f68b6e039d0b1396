"""first-trip probe: timing of the rollout kernel at config 2 / bigger sizes (not a test, not the bench)"""
import sys, time
import torch
import mpc_amd as M
from mpc_amd import workloads as W
s = M.Solver(0)
p = W.config2_problem()
for prec in (M.PREC_FAST, M.PREC_F64):
    for n in (65536, 1 << 20, 1 << 22):
        s.upload([p], [W.SEED])
        for it in range(3):
            s.run_seeded(0, n, prec)
            r = s.fetch()[0]
        steps = n * p.depth * p.n_substeps
        print("prec=%d n=%d kernel_ms=%.3f total_ms=%.3f steps/s=%.3e refined=%d best=%d" % (
            prec, n, r.kernel_ms, r.total_ms, steps / (r.kernel_ms * 1e-3), r.refined, r.best_ordinal), flush=True)
