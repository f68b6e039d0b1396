"""Depth 8 on ONE GPU as sequential shards (python tools/gpu_probe_depth8.py [fast|strict] [n_shards] [split] [scenario]):
the wall time of the n_shards mpc_hip_solve_tree_shard calls + the combine, against the reference's 0.1-s budget."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import mpc_amd as M
from mpc_amd import workloads as W
from mpc_amd.distributed import tree_levels_tensor, combine_tree_levels

mode = sys.argv[1] if len(sys.argv) > 1 else "fast"
n_shards = int(sys.argv[2]) if len(sys.argv) > 2 else 8
split = int(sys.argv[3]) if len(sys.argv) > 3 else 3
scenario = sys.argv[4] if len(sys.argv) > 4 else "config1"
depth = 8
prec = M.PREC_FAST if mode == "fast" else M.PREC_F64
p = W.config1_problem(depth=depth)[0] if scenario == "config1" else W.semicircle_problem(depth + 1.5)[0]
solver = M.Solver(0)
for rep in range(3):
    t0 = time.perf_counter()
    shards = [solver.solve_tree(p, depth, precision=prec, split_level=split, shard=s, n_shards=n_shards) for s in range(n_shards)]
    t1 = time.perf_counter()
    comb = combine_tree_levels(torch.stack([tree_levels_tensor(lv, depth) for _, lv in shards]), split)
    t2 = time.perf_counter()
    total, ordinal, bits, owner = comb[-1]
    print("%s %s depth 8, %d shards split at %d: %.2f ms (+ %.2f ms combine); leaves %d, winner ordinal %d score %.17g; per shard ms: %s"
          % (scenario, mode, n_shards, split, (t1 - t0) * 1e3, (t2 - t1) * 1e3, total, ordinal, np.int64(bits).view(np.float64),
             " ".join("%.1f" % 0 for _ in range(0))), flush=True)
