"""config-2 stream of solves at pipeline levels 0/1/2: time per solve and identical winners (python tools/gpu_probe_pair.py)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, mpc_amd as M
from mpc_amd import workloads as W
depth, count = 20, 65536
stream = torch.cuda.Stream()
with torch.cuda.stream(stream):
    s = M.Solver(0, stream=stream.cuda_stream); s.set_timing(0)
    prob = W.config2_problem(depth)
    r, t = W.seeded_controls(W.SEED, 0, count, depth, prob.last_rudder, prob.last_throttle)
    d_r = torch.tensor(r, dtype=torch.float32, device="cuda"); d_t = torch.tensor(t, dtype=torch.float32, device="cuda")
    d_last = torch.empty(count, dtype=torch.float64, device="cuda"); d_cum = torch.empty(count, dtype=torch.float64, device="cuda")
    s.upload([prob], [W.SEED])
    for level in (0, 1, 2):
        for seeded in (False, True):
            s.set_pipeline(level)
            N = 4000
            sink = torch.zeros((N, 2), dtype=torch.int64, device="cuda")
            def go(n):
                for i in range(n):
                    s.set_winner_sink(sink[i])
                    if seeded: s.run_seeded(0, count, M.PREC_FAST, M.SELECT_LAST_STEP, d_last, d_cum)
                    else: s.run_candidates(d_r, d_t, count, M.PREC_FAST, M.SELECT_LAST_STEP, d_last, d_cum)
                s.flush()
            go(200); torch.cuda.synchronize()
            t0 = time.perf_counter(); go(N); torch.cuda.synchronize(); dt = time.perf_counter() - t0
            w = sink.cpu().numpy()
            print("level %d %s: %.2f us per solve, %.4g steps/s, distinct winners %d %s" % (
                level, "seeded" if seeded else "explicit", dt / N * 1e6, N * count * 400 / dt, len(np.unique(w, axis=0)), w[0]), flush=True)
            # regions of 20 solves, as the driver times them
            regs = []
            for _ in range(50):
                torch.cuda.synchronize(); t0 = time.perf_counter(); go(20); torch.cuda.synchronize(); regs.append(time.perf_counter() - t0)
            print("    20-solve regions: median %.2f us per solve" % (np.median(regs) / 20 * 1e6), flush=True)
    s.set_pipeline(0)
print("ok")
