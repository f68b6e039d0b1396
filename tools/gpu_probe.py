"""perf probe (not a test, not the bench): rollout-kernel time for a library variant / workgroup size.
usage: python tools/gpu_probe.py [lib.so]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: F401  (one HIP runtime per process)
import mpc_amd as M
from mpc_amd import binding as B
from mpc_amd import workloads as W

if len(sys.argv) > 1:
    B._lib = B.load_library(sys.argv[1])
s = M.Solver(0)
p = W.config2_problem()
for prec in (M.PREC_FAST, M.PREC_F64):
    for n in (65536, 1 << 22):
        s.upload([p], [W.SEED])
        for it in range(5):
            s.run_seeded(0, n, prec)
        s.timing(reset=True)
        for it in range(20):
            s.run_seeded(0, n, prec)
        runs, k1, tot = s.timing()
        r = s.fetch()[0]
        steps = n * p.depth * p.n_substeps
        print("%s blk=%s prec=%d n=%d K1_ms=%.4f solve_ms=%.4f K1 steps/s=%.3e best=%d" % (
            os.path.basename(sys.argv[1]) if len(sys.argv) > 1 else "default", os.environ.get("MPC_K1_BLOCK", "-"),
            prec, n, k1 / runs, tot / runs, steps / (k1 / runs * 1e-3), r.best_ordinal), flush=True)
