"""A long random run of the SHARDED tree search (python tools/fuzz_tree_shards.py [problems] [seed]): random problems of depth
4-6, 2-8 shards, split after level 1-3; the shards run one after the other on this GPU and their per-level winners are
combined as distributed.solve_tree_sharded combines them: every level's leaf count, global DFS ordinal and score bits,
and the owning shard's trajectory, must be the whole search's -- in both expansions."""
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
import mpc_amd as M
from mpc_amd import workloads as W
from mpc_amd.distributed import tree_levels_tensor, combine_tree_levels
from helpers import traj_array

n_problems = int(sys.argv[1]) if len(sys.argv) > 1 else 150
rng = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 31)
solver = M.Solver(0)
bad = []
t0 = time.time()
for k in range(n_problems):
    depth = int(rng.randint(4, 7))
    speed = float(rng.uniform(0.5, 2.75))
    words = [(int(rng.randint(0, 4)), (float(rng.uniform(0.1, 1.5)), float(rng.uniform(0.5, 4.0)), float(rng.uniform(0.1, 1.5)))),
             (W.LSL, (0.0, 40.0, 0.0))]
    plan = W.chain(words, float(rng.uniform(6.0, 20.0)), speed, (float(rng.uniform(-100, 100)), float(rng.uniform(-100, 100)),
                                                                  float(rng.uniform(0, 2 * math.pi))), W.T0)
    cps = [plan.sample(W.T0 + (i + 1) * 1.0) for i in range(depth)]
    x, y, heading, _ = plan.sample(W.T0)
    start = (x + float(rng.uniform(-3, 3)), y + float(rng.uniform(-3, 3)), (heading + float(rng.uniform(-0.6, 0.6))) % (2 * math.pi),
             float(rng.uniform(0.0, 3.2)))
    w = rng.uniform(0.5, 30.0, size=3)
    p = M.make_problem(start, cps, M.substep_schedule(1.0), last=(float(rng.randint(-16, 17)) * 0.0625, float(rng.randint(0, 9)) * 0.125),
                       current=(float(rng.uniform(-0.8, 0.8)), float(rng.uniform(-0.8, 0.8))), weights=tuple(float(v) for v in w))
    n_shards, split = int(rng.randint(2, 9)), int(rng.randint(1, 4))
    for prec in (M.PREC_F64, M.PREC_FAST):
        whole, wl = solver.solve_tree(p, depth, precision=prec)
        shards = [solver.solve_tree(p, depth, precision=prec, split_level=split, shard=s, n_shards=n_shards) for s in range(n_shards)]
        combined = combine_tree_levels(torch.stack([tree_levels_tensor(lv, depth) for _, lv in shards]), split)
        why = []
        for l in range(depth):
            total, ordinal, bits, owner = combined[l]
            if total != wl[l].n_leaves or ordinal != wl[l].best_leaf or bits != int(np.float64(wl[l].best_score).view(np.int64)):
                why.append("level %d: (%d, %d) vs (%d, %d)" % (l, total, ordinal, wl[l].n_leaves, wl[l].best_leaf))
        if traj_array(shards[combined[-1][3]][0]).tobytes() != traj_array(whole).tobytes():
            why.append("the owner's trajectory differs")
        if why:
            bad.append((k, prec, depth, n_shards, split, why[:3]))
    if k % 25 == 24:
        print("%d problems, disagreements %d, %.0f s" % (k + 1, len(bad), time.time() - t0), flush=True)
print("sharded tree problems", n_problems, "; disagreements", len(bad))
for b in bad:
    print("   ", b)
