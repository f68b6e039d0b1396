"""A long random run of the EXACT cross-shard decision (python tools/fuzz_shards.py [problems] [seed]): a random problem's
seeded candidate set is cut into 2-8 contiguous shards of random sizes, every shard runs on its own (as a GPU of a node
would), the winner records are stacked as an all-gather would deliver them and mpc_hip_combine_shards_exact decides;
the result must be the winner of the whole set in ONE launch (fast mode: strict-decided) and the oracle's choice."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
import mpc_amd as M
from helpers import oracle_seeded
import test_gpu_random_problems as T


def run(n_problems=300, seed=9, solver=None, verbose=True, sizes=(512, 1000, 4096)):
    """-> dict(problems = those run, bad = [disagreements], tied_rows = rows that had a cross-shard near-tie)"""
    rng = np.random.RandomState(seed)
    solver = solver or M.Solver(0)
    bad = []
    t0 = time.time()
    done = 0
    tied0 = solver.tied_rows()
    for k in range(n_problems):
        p = T._problem(rng)
        if max(p.substeps[i] for i in range(p.n_substeps)) > 0.1000001:
            continue
        seed_k = int(rng.randint(1, 2 ** 31))
        n = int(rng.choice(sizes))
        S = int(rng.randint(2, 9))
        cuts = np.sort(rng.choice(np.arange(1, n), size=S - 1, replace=False))
        bounds = [0] + [int(c) for c in cuts] + [n]
        whole = solver.eval_seeded(p, seed_k, 0, n, M.PREC_FAST)
        _, _, o_best = oracle_seeded(p, seed_k, 0, n)
        W = M.winner_record_words(p.depth)
        recs = torch.zeros((S, 1, W), dtype=torch.int64, device="cuda")
        torch.cuda.synchronize()
        solver.upload([p], [seed_k])
        for s in range(S):
            solver.run_seeded(bounds[s], bounds[s + 1] - bounds[s], M.PREC_FAST)        # ordinals are global: stride 0
            solver.copy_winner_records(recs[s], p.depth)
        out = torch.zeros((1, 2), dtype=torch.int64, device="cuda")
        solver.combine_shards_exact(recs, S, 1, p.depth, 0, out, M.PREC_FAST, M.SELECT_LAST_STEP)
        solver.synchronize()
        got = int(out[0, 1])
        done += 1
        if got != int(whole.best_ordinal) or (got != int(o_best.best_index) and not whole.libm_tie):
            bad.append((k, S, n, p.depth, p.n_substeps, got, int(whole.best_ordinal), int(o_best.best_index)))
        if verbose and k % 50 == 49:
            print("%d problems, disagreements %d, rows with a cross-shard near-tie so far %d, %.0f s" % (done, len(bad), solver.tied_rows() - tied0, time.time() - t0), flush=True)
    return {"problems": done, "bad": bad, "tied_rows": solver.tied_rows() - tied0}


if __name__ == "__main__":
    out = run(int(sys.argv[1]) if len(sys.argv) > 1 else 300, int(sys.argv[2]) if len(sys.argv) > 2 else 9)
    print("sharded problems", out["problems"], "; disagreements", len(out["bad"]), "; rows that had a cross-shard near-tie", out["tied_rows"])
    for b in out["bad"]:
        print("   ", b)
