"""A long random run of FLEETS against the oracle (python tools/fuzz_fleet.py [fleets] [seed]): 2-40 random problems per batch,
every one with its own depth, substep schedule, weights and current (tests/test_gpu_random_problems.py's generator), one
seeded candidate set size per batch; both precisions; pipeline levels 0, 1, 2 must agree bit for bit."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
import mpc_amd as M
from helpers import oracle_seeded, traj_array
import test_gpu_random_problems as T

n_fleets = int(sys.argv[1]) if len(sys.argv) > 1 else 100
rng = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 5)
solver = M.Solver(0)
bad, wrong, checked = [], 0, 0
worst = {M.PREC_F64: 0.0, M.PREC_FAST: 0.0}
t0 = time.time()
for k in range(n_fleets):
    P = int(rng.randint(2, 41))
    probs = [T._problem(rng) for _ in range(P)]
    seeds = [int(rng.randint(1, 2 ** 31)) for _ in range(P)]
    n = int(rng.choice([64, 100, 256, 1000, 2048]))
    oracle = [oracle_seeded(p, s, 0, n) for p, s in zip(probs, seeds)]
    for prec in (M.PREC_F64, M.PREC_FAST):
        if prec == M.PREC_FAST and any(max(p.substeps[i] for i in range(p.n_substeps)) > 0.1000001 for p in probs):
            continue
        seen = []
        for level in (0, 1, 2):
            solver.set_pipeline(level)
            solver.upload(probs, seeds)
            last = torch.empty(P * n, dtype=torch.float64, device="cuda")
            solver.run_seeded(0, n, prec, M.SELECT_LAST_STEP, last, None)
            res = solver.fetch()
            seen.append((last.cpu().numpy().tobytes(), [(r.status, r.best_ordinal, r.best_last, traj_array(r).tobytes()) for r in res]))
        solver.set_pipeline(0)
        if not (seen[0] == seen[1] == seen[2]):
            bad.append((k, prec, P, n, "pipeline levels disagree"))
        g = np.frombuffer(seen[0][0], dtype=np.float64).reshape(P, n)
        for q, (p, (o_last, o_cum, o_best)) in enumerate(zip(probs, oracle)):
            floor = 0.01 * (p.w_distance + p.w_heading + p.w_speed)
            ok = np.isfinite(o_last) & np.isfinite(g[q])
            e = float((np.abs(g[q][ok] - o_last[ok]) / np.maximum(np.abs(o_last[ok]), floor)).max()) if ok.any() else 0.0
            worst[prec] = max(worst[prec], e)
            checked += 1
            st, bo, bl, _ = seen[0][1][q]
            if bo != o_best.best_index or seen[0][1][q][0] != 0:
                wrong += 1
                bad.append((k, prec, P, n, q, p.depth, p.n_substeps, "choice", int(bo), int(o_best.best_index), st))
            elif e > 10 * T.TOL[prec]:
                bad.append((k, prec, P, n, q, p.depth, p.n_substeps, "error", e))
    if k % 20 == 19:
        print("%d fleets (%d problem solves), wrong choices %d, findings %d, worst strict %.3g fast %.3g, %.0f s" % (
            k + 1, checked, wrong, len(bad), worst[M.PREC_F64], worst[M.PREC_FAST], time.time() - t0), flush=True)
print("fleets:", n_fleets, "; problem solves checked", checked, "; wrong choices", wrong,
      "; worst error relative to max(|cost|, 1 %% of the weights' sum): strict %.3g fast %.3g" % (worst[M.PREC_F64], worst[M.PREC_FAST]))
for b in bad:
    print("   ", b)
