"""perf probe (not a test): the reference-native tree at depth 8 -- 5.1e9 leaves -- as the 8 shards of a node's 8 GPUs,
run one after the other on this GPU (mpc_hip_solve_tree_shard), plus the combine."""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: F401
import mpc_amd as M
from mpc_amd import workloads as W
from mpc_amd.distributed import tree_levels_tensor, combine_tree_levels

prec = M.PREC_F64 if len(sys.argv) > 1 and sys.argv[1] == "f64" else M.PREC_FAST
depth, n_shards, split = 8, 8, 2
p, _ = W.config1_problem(depth=depth)
s = M.Solver(0)
rows = []
for shard in range(n_shards):
    t0 = time.perf_counter()
    res, lv = s.solve_tree(p, depth, precision=prec, split_level=split, shard=shard, n_shards=n_shards)
    wall = (time.perf_counter() - t0) * 1e3
    print("shard %d: levels=%d leaves(last)=%d evaluated=%d gpu_ms=%.2f wall_ms=%.2f steps/s=%.3e" % (
        shard, res.levels, int(lv[res.levels - 1].n_leaves), res.evaluated, res.total_ms, wall,
        res.rollout_steps / (res.total_ms * 1e-3)), flush=True)
    rows.append(tree_levels_tensor(lv, res.levels))
comb = combine_tree_levels(torch.stack(rows), split)
for l, c in enumerate(comb):
    print("level %d: leaves=%d winner=%d owner=%d" % (l + 1, c[0], c[1], c[3]))
