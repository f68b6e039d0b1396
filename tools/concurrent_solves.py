"""Throughput of independent config-2 solves issued round-robin to S handles on S streams (usage: concurrent_solves.py S [out] [n]);
MPC_PIPELINE=1 in the environment: every handle with the deferred finish."""
import os, sys, time
sys.path.insert(0, ".")
import torch, mpc_amd as M
from mpc_amd import workloads as W
S = int(sys.argv[1]) if len(sys.argv) > 1 else 2
depth, count = 20, 65536
prob = W.config2_problem(depth)
r, t = W.seeded_controls(W.SEED, 0, count, depth, prob.last_rudder, prob.last_throttle)
d_rud = torch.tensor(r, dtype=torch.float32, device="cuda").contiguous()
d_thr = torch.tensor(t, dtype=torch.float32, device="cuda").contiguous()
streams = [torch.cuda.Stream() for _ in range(S)]
solvers = []
for st in streams:
    s = M.Solver(0, stream=st.cuda_stream); s.upload([prob], [W.SEED]); s.set_timing(0); s.set_pipeline(os.environ.get('MPC_PIPELINE') == '1'); solvers.append(s)
torch.cuda.synchronize()
OUT = len(sys.argv) > 2
d_last = [torch.empty(count, dtype=torch.float64, device="cuda") for _ in range(S)]
d_cum = [torch.empty(count, dtype=torch.float64, device="cuda") for _ in range(S)]
def run(n):
    for i in range(n):
        if OUT: solvers[i % S].run_candidates(d_rud, d_thr, count, M.PREC_FAST, M.SELECT_LAST_STEP, d_last[i % S], d_cum[i % S])
        else: solvers[i % S].run_candidates(d_rud, d_thr, count, M.PREC_FAST, M.SELECT_LAST_STEP)
run(200); torch.cuda.synchronize()
n = int(sys.argv[3]) if len(sys.argv) > 3 else 4000
t0 = time.perf_counter(); run(n); [s.flush() for s in solvers]; torch.cuda.synchronize(); dt = time.perf_counter() - t0
print("streams %d: %.4f ms per solve, %.4g steps/s" % (S, dt / n * 1e3, n * count * 400 / dt))
