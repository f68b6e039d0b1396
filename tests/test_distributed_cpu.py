"""The multi-GPU combine step (mpc_amd/distributed.py) on CPU: world_size 2 and 3 over gloo.  The data path needs
no other collective: shards are contiguous ordinal ranges of one candidate set."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from mpc_amd.distributed import combine_winners, shard_range, owner_rank, bits_to_score


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _bits(x):
    return int(np.array([x], dtype=np.float64).view(np.int64)[0])


def _worker(rank, world, port, scores, ordinals, n_total, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    first, count = shard_range(n_total, rank, world)
    P = len(scores[rank])
    w = torch.tensor([[_bits(scores[rank][p]), ordinals[rank][p]] for p in range(P)], dtype=torch.int64)
    key, cand = combine_winners(w, ordinal_offset=first)
    out[rank] = (bits_to_score(key).tolist(), cand.tolist())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_min_allreduce_combine(world):
    n_total = 1000
    # problem 0: distinct minima; problem 1: an exact tie between ranks 0 and last -> lowest GLOBAL ordinal wins;
    # problem 2: the minimum is on the last rank; problem 3: +inf everywhere
    scores = [[3.5, 2.0, 9.0, float("inf")] for _ in range(world)]
    ordinals = [[7, 11, 0, 5] for _ in range(world)]
    scores[0][0] = 1.25
    scores[-1][2] = 0.5
    with mp.Manager() as m:
        out = m.dict()
        mp.spawn(_worker, args=(world, _free_port(), scores, ordinals, n_total, out), nprocs=world, join=True)
        res = [out[r] for r in range(world)]
    assert all(r == res[0] for r in res)                      # identical on every rank
    key, cand = res[0]
    last_first = shard_range(n_total, world - 1, world)[0]
    assert key[:3] == [1.25, 2.0, 0.5] and key[3] == float("inf")
    assert cand[0] == 7 and cand[1] == 11 and cand[2] == last_first + 0 and cand[3] == 5
    assert owner_rank(cand[2], n_total, world) == world - 1 and owner_rank(cand[0], n_total, world) == 0


def test_shard_ranges_partition_the_set():
    for n, world in [(65536, 8), (16777216, 8), (1000, 3), (5, 8), (7, 1)]:
        ranges = [shard_range(n, r, world) for r in range(world)]
        assert ranges[0][0] == 0 and sum(c for _, c in ranges) == n
        for r in range(1, world):
            assert ranges[r][0] == ranges[r - 1][0] + ranges[r - 1][1]
        for g in (0, n // 2, n - 1):
            r = owner_rank(g, n, world)
            assert ranges[r][0] <= g < ranges[r][0] + ranges[r][1]


def test_single_process_combine_is_the_identity():
    w = torch.tensor([[_bits(2.5), 42]], dtype=torch.int64)
    key, cand = combine_winners(w, ordinal_offset=100)
    assert bits_to_score(key)[0] == 2.5 and cand.tolist() == [142]


def _gather_worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from mpc_amd.distributed import combine_gathered
    P = 3
    # per-shard local winners: problem 0 best on rank 1, problem 1 tie (lowest global ordinal wins), problem 2 empty (-1)
    score = [[5.0, 2.0, float("inf")], [1.5, 2.0, float("inf")]][rank]
    ordl = [[9, 4, -1], [3, 4, -1]][rank]
    w = torch.tensor([[_bits(score[p]), ordl[p]] for p in range(P)], dtype=torch.int64).view(-1)
    g = torch.zeros(world * P * 2, dtype=torch.int64)
    dist.all_gather_into_tensor(g, w)
    key, cand = combine_gathered(g, world, P, ordinal_stride=100)
    out[rank] = (bits_to_score(key).tolist(), cand.tolist())
    dist.barrier()
    dist.destroy_process_group()


def test_all_gather_combine_matches_the_two_allreduce_form():
    world = 2
    with mp.Manager() as m:
        out = m.dict()
        mp.spawn(_gather_worker, args=(world, _free_port(), out), nprocs=world, join=True)
        res = [out[r] for r in range(world)]
    assert res[0] == res[1]
    key, cand = res[0]
    assert key[:2] == [1.5, 2.0] and key[2] == float("inf")
    assert cand[0] == 100 + 3 and cand[1] == 4 and cand[2] == np.iinfo(np.int64).max


class _FakeLevel:
    def __init__(self, n_leaves, best_leaf, best_score):
        self.n_leaves, self.best_leaf, self.best_score = n_leaves, best_leaf, best_score


class _FakeResult:
    def __init__(self, levels):
        self.levels = levels


class _FakeSolver:
    """stands in for the GPU solver: what rank r's shard of a search would report per level"""

    def __init__(self, table):
        self.table = table

    def solve_tree(self, problem, max_levels, precision, split_level, shard, n_shards):
        lv = [_FakeLevel(*self.table[shard][l]) for l in range(max_levels)]
        return _FakeResult(max_levels), lv


def _tree_worker(rank, world, port, table, out):
    from mpc_amd.distributed import solve_tree_sharded
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    combined, res, owner = solve_tree_sharded(_FakeSolver(table), None, 4, split_level=2)
    out[rank] = (combined, owner)
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_tree_levels_combine_over_gloo():
    world = 3
    # levels 0, 1: expanded in full by every shard; level 2: the minimum sits on shard 1; level 3: an exact tie between
    # shards 0 and 2 -> the lower global DFS ordinal (shard 0's) wins
    table = [
        [(15, 4, 2.5), (270, 33, 1.5), (1000, 999, 0.75), (18000, 17, 0.125)],
        [(15, 4, 2.5), (270, 33, 1.5), (1200, 5, 0.5), (20000, 0, 0.25)],
        [(15, 4, 2.5), (270, 33, 1.5), (900, 0, 0.75), (16000, 3, 0.125)],
    ]
    with mp.Manager() as m:
        out = m.dict()
        mp.spawn(_tree_worker, args=(world, _free_port(), table, out), nprocs=world, join=True)
        res = [out[r] for r in range(world)]
    assert res[0] == res[1] == res[2]
    combined, owner = res[0]
    assert combined[0] == (15, 4, _bits(2.5), 0) and combined[1] == (270, 33, _bits(1.5), 0)
    assert combined[2] == (3100, 1000 + 5, _bits(0.5), 1)
    assert combined[3] == (54000, 17, _bits(0.125), 0) and owner == 0


def _pieces_worker(rank, world, port, table, out):
    from mpc_amd.distributed import solve_tree_sharded
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    solver = _FakeSolver(table)
    asked = []
    plain = solver.solve_tree
    def spy(problem, max_levels, precision, split_level, shard, n_shards):
        asked.append((shard, n_shards))
        return plain(problem, max_levels, precision, split_level, shard, n_shards)
    solver.solve_tree = spy
    combined, res, owner = solve_tree_sharded(solver, None, 4, split_level=2, pieces_per_rank=2)
    out[rank] = (combined, owner, asked)
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_tree_with_several_pieces_per_rank():
    """a level too wide even for a world-th of it (depth 9 on a node): world x pieces shards, a rank runs its consecutive
    pieces one after the other; the combination is that of the 4 shards in slice order, the owner the rank that ran the piece"""
    from mpc_amd.distributed import combine_tree_levels, tree_levels_tensor
    table = [
        [(15, 4, 2.5), (270, 33, 1.5), (1000, 999, 0.75), (18000, 17, 0.25)],
        [(15, 4, 2.5), (270, 33, 1.5), (1200, 5, 0.5), (20000, 0, 0.25)],
        [(15, 4, 2.5), (270, 33, 1.5), (900, 0, 0.75), (16000, 3, 0.125)],
        [(15, 4, 2.5), (270, 33, 1.5), (800, 7, 0.5), (15000, 9, 0.125)],
    ]
    with mp.Manager() as m:
        out = m.dict()
        mp.spawn(_pieces_worker, args=(2, _free_port(), table, out), nprocs=2, join=True)
        res = [out[r] for r in range(2)]
    assert res[0][:2] == res[1][:2]
    assert res[0][2] == [(0, 4), (1, 4)] and res[1][2] == [(2, 4), (3, 4)]
    combined, owner, _ = res[0]
    want = combine_tree_levels(torch.stack([tree_levels_tensor([_FakeLevel(*t) for t in table[s]], 4) for s in range(4)]), 2)
    assert combined == want
    assert combined[2] == (3900, 1000 + 5, _bits(0.5), 1) and combined[3] == (69000, 18000 + 20000 + 3, _bits(0.125), 2) and owner == 1


def _exact_worker(rank, world, port, out):
    from mpc_amd.distributed import combine_winners_exact
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    # problem 0: the shards' fast scores differ by 3e-8 relative (inside the fast model's error), and the strict model
    # ranks them the other way round; problem 1: clearly apart; problem 2: no finite score anywhere
    fast = [[2.00000000, 5.0, float("inf")], [2.00000006, 4.0, float("inf")]][rank]
    strict = {0: [2.0000000300, 2.0000000299][rank]}
    local = [[3, 8, -1], [5, 2, -1]][rank]
    calls = []

    def strict_score(p, ordinal):
        calls.append((p, ordinal))
        return strict[p]
    w = torch.tensor([[_bits(fast[p]), local[p]] for p in range(3)], dtype=torch.int64)
    key, cand = combine_winners_exact(w, ordinal_offset=rank * 100, strict_score=strict_score)
    out[rank] = (bits_to_score(key).tolist(), cand.tolist(), calls)
    dist.barrier()
    dist.destroy_process_group()


def test_cross_shard_near_ties_are_decided_by_the_strict_model():
    with mp.Manager() as m:
        out = m.dict()
        mp.spawn(_exact_worker, args=(2, _free_port(), out), nprocs=2, join=True)
        res = [out[r] for r in range(2)]
    assert res[0][:2] == res[1][:2]
    key, cand, _ = res[0]
    assert key[0] == 2.0000000299 and cand[0] == 100 + 5          # rank 1's winner, although its fast score was higher
    assert key[1] == 4.0 and cand[1] == 100 + 2
    assert key[2] == float("inf") and cand[2] == np.iinfo(np.int64).max
    assert res[0][2] == [(0, 3)] and res[1][2] == [(0, 5)]         # only the tied problem was re-scored, once per shard


class _RaggedSolver(_FakeSolver):
    """shards that stopped at different depths (a node budget, the storage cap or memory can do that: they expand
    slices of different widths)"""

    def __init__(self, table, completed):
        super().__init__(table)
        self.completed = completed

    def solve_tree(self, problem, max_levels, precision, split_level, shard, n_shards):
        n = self.completed[shard]
        return _FakeResult(n), [_FakeLevel(*self.table[shard][l]) for l in range(n)]


def _ragged_worker(rank, world, port, table, completed, out):
    from mpc_amd.distributed import solve_tree_sharded
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    combined, res, owner = solve_tree_sharded(_RaggedSolver(table, completed), None, 4, split_level=2)
    out[rank] = (combined, owner)
    dist.barrier()
    dist.destroy_process_group()


def test_shards_that_stop_at_different_depths_combine_their_common_levels():
    """advisor (round 1): the all-gather needs equal shapes, and only the levels EVERY shard completed have a
    combined winner; a shard that selected nothing at a level (no finite score: ordinal -1) does not take part"""
    world = 2
    none = int(np.uint64(0xFFFFFFFFFFFFFFFF).view(np.int64))
    table = [
        [(15, 4, 2.5), (270, 33, 1.5), (1000, none & 0xFFFFFFFFFFFFFFFF, float("inf")), (18000, 17, 0.125)],
        [(15, 4, 2.5), (270, 33, 1.5), (1200, 5, 0.5), (20000, 0, 0.25)],
    ]
    with mp.Manager() as m:
        out = m.dict()
        mp.spawn(_ragged_worker, args=(world, _free_port(), table, [4, 3], out), nprocs=world, join=True)
        res = [out[r] for r in range(world)]
    assert res[0] == res[1]
    combined, owner = res[0]
    assert len(combined) == 3                                              # shard 1 completed only three levels
    assert combined[2] == (2200, 1000 + 5, _bits(0.5), 1) and owner == 1   # shard 0 selected nothing at level 3
