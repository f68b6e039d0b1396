"""One rank of tests/test_gpu_two_ranks.py: a fresh process (started by the test with subprocess, never a re-exec of a
process that has touched the GPU), process group gloo, BOTH ranks on cuda:0 -- RCCL refuses two ranks on one device,
gloo does not care -- so that the world-size-2 code of mpc_amd/distributed.py runs end to end with real kernels on
every rank on a 1-GPU box.  Writes what it saw as JSON; the test compares the ranks and asserts.

    RANK=r WORLD_SIZE=2 MASTER_ADDR=127.0.0.1 MASTER_PORT=p python tests/two_rank_worker.py shards|flip|tree out.json
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

N_SHARD = 65536
DEPTH = 20


def bits(x):
    return int(np.float64(x).view(np.int64))


def run_shards(torch, dist, M, W, solver, rank, world):
    """Each rank: its 65,536-candidate shard of ONE 131,072-candidate set of BASELINE config 2, solves streamed through
    ShardCombiner(exact=True) at every pipeline level; then the whole set as one launch, and (rank 0) the oracle."""
    from mpc_amd.distributed import ShardCombiner
    p = W.config2_problem(DEPTH)
    first = rank * N_SHARD
    rud, thr = W.seeded_controls(W.SEED, first, N_SHARD, DEPTH, p.last_rudder, p.last_throttle)
    d_rud = torch.tensor(rud, dtype=torch.float32, device="cuda").contiguous()
    d_thr = torch.tensor(thr, dtype=torch.float32, device="cuda").contiguous()
    solver.upload([p], [W.SEED])
    out = {"combined": {}}
    for level in (0, 1, 2):
        solver.set_pipeline(level)
        comb = ShardCombiner(solver, 1, ordinal_stride=N_SHARD, batch=4, exact=True, depth=DEPTH)
        assert comb.collective and comb.host_staged and comb.world == world
        def solves(n):
            for _ in range(n):
                comb.begin()
                solver.run_candidates(d_rud, d_thr, N_SHARD, M.PREC_FAST)
                comb.submit()
            k, o = comb.flush()
            torch.cuda.synchronize()
            return int(k[0].item()), int(o[0].item())
        full = solves(8)                         # two full batches: every row of the last one decided alike
        rows = comb.results.view(comb.B, 2).cpu().numpy()
        alike = bool((rows == rows[0]).all())
        part = solves(1)                         # a partial batch in a reused slot: the rows behind it read "nothing selected"
        rows = comb.results.view(comb.B, 2).cpu().numpy()
        empty_tail = bool((rows[1:, 1] == -1).all())
        out["combined"][str(level)] = {"bits": full[0], "ordinal": full[1], "rows_alike": alike, "partial": list(part),
                                       "empty_tail": empty_tail, "tied_rows": comb.tied_rows()}
    solver.set_pipeline(0)
    mine = solver.fetch()[0]
    out["shard"] = {"bits": bits(mine.best_last), "ordinal": int(mine.best_ordinal) + first}
    # the whole set in ONE launch on this rank's own handle
    rud2, thr2 = W.seeded_controls(W.SEED, 0, world * N_SHARD, DEPTH, p.last_rudder, p.last_throttle)
    whole = M.Solver(0)
    r = whole.eval_candidates(p, torch.tensor(rud2, dtype=torch.float32, device="cuda").contiguous(),
                              torch.tensor(thr2, dtype=torch.float32, device="cuda").contiguous(), world * N_SHARD, M.PREC_FAST)
    out["one_launch"] = {"bits": bits(r.best_last), "ordinal": int(r.best_ordinal), "rudder": r.first_rudder, "throttle": r.first_throttle}
    whole.close()
    if rank == 0:
        from helpers import oracle_seeded
        _, _, best = oracle_seeded(p, W.SEED, 0, world * N_SHARD)
        out["oracle"] = {"ordinal": int(best.best_index), "last": float(best.best_last)}
        out["one_launch"]["last"] = float(r.best_last)
    return out


def run_flip(torch, dist, M, W, solver, rank, world):
    """The constructed pair of tests/test_gpu_full_size.py (adjacent in strict order, inverted in fast order), one
    candidate per rank, in both rank orders: the fast-key combine prefers the wrong rank, the exact one the right one."""
    from mpc_amd.distributed import ShardCombiner
    p = W.config2_problem(DEPTH)
    n = 1 << 20
    d = torch.empty(n, dtype=torch.float64, device="cuda")
    solver.eval_seeded(p, W.SEED, 0, n, M.PREC_FAST, M.SELECT_LAST_STEP, d, None)
    fast = d.cpu().numpy().copy()
    solver.eval_seeded(p, W.SEED, 0, n, M.PREC_F64, M.SELECT_LAST_STEP, d, None)
    strict = d.cpu().numpy().copy()
    order = np.argsort(strict, kind="stable")
    fs, ss = fast[order], strict[order]
    flips = np.flatnonzero((ss[1:] > ss[:-1]) & (fs[1:] < fs[:-1]) & np.isfinite(ss[1:]))
    assert flips.size > 0
    a, b = int(order[flips[0]]), int(order[flips[0] + 1])                  # strict: a < b; fast: b < a
    assert strict[a] < strict[b] and fast[b] < fast[a] <= fast[b] * (1 + M.REFINE_REL)
    out = {"a": a, "b": b, "strict_a_bits": bits(strict[a]), "fast_b_bits": bits(fast[b]), "rounds": []}
    for holders in ((b, a), (a, b)):                                        # holders[r] = the candidate rank r holds
        o = holders[rank]
        r_, t_ = W.seeded_controls(W.SEED, o, 1, DEPTH, p.last_rudder, p.last_throttle)
        d_r = torch.tensor(r_.astype(np.float32), device="cuda").contiguous()
        d_t = torch.tensor(t_.astype(np.float32), device="cuda").contiguous()
        solver.upload([p], [W.SEED])
        got = {}
        for exact in (False, True):
            comb = ShardCombiner(solver, 1, ordinal_stride=1, batch=2, exact=exact, depth=DEPTH)     # stride 1: global ordinal = rank
            comb.begin()
            solver.run_candidates(d_r, d_t, 1, M.PREC_FAST)
            comb.submit()
            k, g = comb.flush()
            torch.cuda.synchronize()
            got["exact" if exact else "fast"] = {"bits": int(k[0].item()), "rank": int(g[0].item())}
        got["rank_of_a"] = holders.index(a)
        got["tied_rows"] = solver.tied_rows()
        out["rounds"].append(got)
    return out


def run_tree(torch, dist, M, W, solver, rank, world):
    """depth 8 (5.2e9 leaves) as solve_tree_sharded runs it: one shard per rank, and again with two pieces per rank; the
    combined record against mpc_hip_solve_tree_deep on this rank's own handle"""
    from mpc_amd.distributed import solve_tree_sharded
    depth = 8
    p = W.config1_problem(depth=depth)[0]
    deep, dl = solver.solve_tree_deep(p, depth, precision=M.PREC_FAST)
    out = {"deep": {"levels": int(deep.levels), "leaves": int(dl[depth - 1].n_leaves), "ordinal": int(dl[depth - 1].best_leaf),
                    "bits": bits(dl[depth - 1].best_score)}}
    for ppr in (1, 2):
        combined, res, owner = solve_tree_sharded(solver, p, depth, split_level=3, precision=M.PREC_FAST, pieces_per_rank=ppr)
        total, ordinal, kbits, shard = combined[-1]
        out["ppr%d" % ppr] = {"levels": len(combined), "leaves": int(total), "ordinal": int(ordinal), "bits": int(kbits), "owner": int(owner),
                              "mine_is_winner": bool(owner == rank and bits(res.best_last) == int(kbits))}
    return out


def main():
    import faulthandler
    faulthandler.enable()                           # a crash in native code names the Python line it came from
    case, out_path = sys.argv[1], sys.argv[2]
    import torch
    import torch.distributed as dist
    import mpc_amd as M
    from mpc_amd import workloads as W
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    assert torch.cuda.is_available()
    torch.cuda.set_device(0)
    dist.init_process_group("gloo")
    stream = torch.cuda.Stream()
    with torch.cuda.stream(stream):                 # the combiner's staging copies run on torch's current stream: the solver's
        solver = M.Solver(0, stream=stream.cuda_stream)
        out = {"shards": run_shards, "flip": run_flip, "tree": run_tree}[case](torch, dist, M, W, solver, rank, world)
        solver.close()
    out["rank"], out["world"] = rank, dist.get_world_size()
    with open(out_path, "w") as f:
        json.dump(out, f)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
