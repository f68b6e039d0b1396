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
