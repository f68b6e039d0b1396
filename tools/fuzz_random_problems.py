"""A long run of tests/test_gpu_random_problems.py's generator (python tools/fuzz_random_problems.py [blocks] [first block]):
every candidate's cost against the oracle in both precisions and the same chosen control, over blocks x 10 random
problems; prints the worst errors and every mismatch instead of stopping at the first."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
import mpc_amd as M
from helpers import oracle_seeded, rel_err
import test_gpu_random_problems as T

blocks = int(sys.argv[1]) if len(sys.argv) > 1 else 100
first_block = int(sys.argv[2]) if len(sys.argv) > 2 else 100
solver = M.Solver(0)
worst = {M.PREC_F64: 0.0, M.PREC_FAST: 0.0}
# the same error against max(|cost|, 1 % of the sum of the weights): a cost whose only weighted term cancels to 1e-4 of its
# operands (say |reference speed - speed over ground| with the other two weights zero) has no relative accuracy to speak of
worst_floored = {M.PREC_F64: 0.0, M.PREC_FAST: 0.0}
wrong_choice = 0
bad = []
n = 1024
t0 = time.time()
done = 0
for block in range(first_block, first_block + blocks):
    rng = np.random.RandomState(1000 + block)
    for k in range(10):
        p = T._problem(rng)
        seed = int(rng.randint(1, 2 ** 31))
        first = int(rng.randint(0, 5000))
        o_last, o_cum, o_best = oracle_seeded(p, seed, first, n)
        for prec in (M.PREC_F64, M.PREC_FAST):
            if prec == M.PREC_FAST and max(p.substeps[i] for i in range(p.n_substeps)) > 0.1000001:
                continue
            last = torch.empty(n, dtype=torch.float64, device="cuda")
            cum = torch.empty(n, dtype=torch.float64, device="cuda")
            res = solver.eval_seeded(p, seed, first, n, prec, M.SELECT_LAST_STEP, last, cum)
            g_last, g_cum = last.cpu().numpy(), cum.cpu().numpy()
            e = max(rel_err(g_last, o_last).max(), rel_err(g_cum, o_cum).max())
            worst[prec] = max(worst[prec], float(e))
            floor = 0.01 * (p.w_distance + p.w_heading + p.w_speed)
            ok = np.isfinite(o_last) & np.isfinite(g_last)
            ef = float((np.abs(g_last[ok] - o_last[ok]) / np.maximum(np.abs(o_last[ok]), floor)).max()) if ok.any() else 0.0
            worst_floored[prec] = max(worst_floored[prec], ef)
            wrong_choice += int(res.best_ordinal != o_best.best_index)
            if e > T.TOL[prec] or res.best_ordinal != o_best.best_index or res.levels != p.depth:
                bad.append((block, k, prec, p.depth, p.n_substeps, float(e), int(res.best_ordinal), int(o_best.best_index)))
        done += 1
    if block % 10 == 9:
        print("block %d: %d problems, worst strict %.3g fast %.3g, mismatches %d, %.0f s" % (
            block, done, worst[M.PREC_F64], worst[M.PREC_FAST], len(bad), time.time() - t0), flush=True)
print("problems", done, "x", n, "candidates; wrong choices:", wrong_choice)
print("worst relative cost error: strict %.3g, fast %.3g" % (worst[M.PREC_F64], worst[M.PREC_FAST]))
print("worst error relative to max(|cost|, 1 %% of the weights' sum): strict %.3g, fast %.3g" % (worst_floored[M.PREC_F64], worst_floored[M.PREC_FAST]))
print("beyond the plain relative tolerance (block, k, precision, depth, substeps, error, chosen, oracle's):")
for b in bad:
    print("   ", b)
