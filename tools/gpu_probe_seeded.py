"""perf probe (not a test): a stream of 65,536-candidate solves with SEEDED (device-generated) controls against the same
set as explicit arrays, two kernels per solve and with the deferred finish."""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import mpc_amd as M
from mpc_amd import workloads as W

p = W.config2_problem()
n = 65536
s = M.Solver(0)
s.set_timing(0)
rud, thr = W.seeded_controls(W.SEED, 0, n, 20, p.last_rudder, p.last_throttle)
d_r = torch.tensor(rud, dtype=torch.float32, device="cuda").contiguous()
d_t = torch.tensor(thr, dtype=torch.float32, device="cuda").contiguous()
s.upload([p], [W.SEED])
for pipeline in (False, True):
    s.set_pipeline(pipeline)
    for name, run in (("seeded  ", lambda: s.run_seeded(0, n, M.PREC_FAST)), ("explicit", lambda: s.run_candidates(d_r, d_t, n, M.PREC_FAST))):
        for _ in range(3000):
            run()
        s.synchronize()
        t0 = time.perf_counter()
        for _ in range(4000):
            run()
        s.synchronize()
        dt = (time.perf_counter() - t0) / 4000
        r = s.fetch()[0]
        print("%s %s: %.2f us per solve, %.3e steps/s, winner %d %.17g" % ("pipelined" if pipeline else "2 kernels ", name, dt * 1e6,
                                                                         n * 400 / dt, r.best_ordinal, r.best_last), flush=True)
