"""BASELINE.json's configurations at their FULL sizes on the GPU.

Config 2 and 3 are small enough for the CPU oracle to score every candidate (about a second each).  Config 4 (one
GPU's 2^21-candidate shard of the 16 M set, 30-s horizon) and config 5 (4096 vessels x 8192 candidates) are checked
through a strided oracle sample plus size-independent properties: the device argmin is the first minimum of the
device costs, the winner's oracle score is <= every sampled oracle score, shards of a set combine to the whole set's
winner, and a problem evaluated alone equals the same problem inside the batch bit for bit.
"""
import numpy as np
import pytest

import mpc_amd as M
from mpc_amd import workloads as W
from mpc_amd.distributed import shard_range
from helpers import oracle_seeded, rel_err

pytestmark = pytest.mark.gpu
TOL_FAST, TOL_F64 = 1e-5, 1e-12


def _torch():
    import torch
    assert torch.cuda.is_available()
    return torch


def _costs(solver, p, seed, first, count, prec):
    torch = _torch()
    last = torch.empty(count, dtype=torch.float64, device="cuda")
    cum = torch.empty(count, dtype=torch.float64, device="cuda")
    res = solver.eval_seeded(p, seed, first, count, prec, M.SELECT_LAST_STEP, last, cum)
    return res, last.cpu().numpy(), cum.cpu().numpy()


@pytest.mark.parametrize("cfg,prec,tol", [("config2", M.PREC_FAST, TOL_FAST), ("config3", M.PREC_F64, TOL_F64)])
def test_config2_and_3_every_candidate(solver, cfg, prec, tol):
    p = W.config2_problem() if cfg == "config2" else W.config3_problem()
    n = 65536
    res, last, cum = _costs(solver, p, W.SEED, 0, n, prec)
    o_last, o_cum, o_best = oracle_seeded(p, W.SEED, 0, n)
    e1, e2 = rel_err(last, o_last).max(), rel_err(cum, o_cum).max()
    print("%s full size: max rel err last %.3e cum %.3e" % (cfg, e1, e2))
    assert e1 <= tol and e2 <= tol
    assert res.best_ordinal == o_best.best_index
    assert (res.first_rudder, res.first_throttle) == tuple(
        a[0, 0] for a in W.seeded_controls(W.SEED, o_best.best_index, 1, 1, p.last_rudder, p.last_throttle))
    assert res.rollout_steps == 65536 * 20 * 20


def test_config4_shard_of_the_16M_set(solver):
    p = W.config2_problem(depth=30)
    total, world = 1 << 24, 8
    first, count = shard_range(total, 3, world)            # rank 3's shard: ordinals 6291456 .. 8388607
    assert count == 1 << 21
    res, last, cum = _costs(solver, p, W.SEED, first, count, M.PREC_FAST)
    stride = 4096
    o_last, o_cum, _ = oracle_seeded(p, W.SEED, first, count // stride, stride=stride)
    assert rel_err(last[::stride], o_last).max() <= TOL_FAST
    assert rel_err(cum[::stride], o_cum).max() <= TOL_FAST
    assert res.best_ordinal == first + int(np.argmin(last))
    w_last, _, _ = oracle_seeded(p, W.SEED, int(res.best_ordinal), 1)
    assert rel_err(res.best_last, w_last[0]) <= TOL_FAST and w_last[0] <= o_last.min()
    # two half-shards combine to the shard's winner (what the min-allreduce does across GPUs)
    half = count // 2
    ra = solver.eval_seeded(p, W.SEED, first, half, M.PREC_FAST)
    rb = solver.eval_seeded(p, W.SEED, first + half, half, M.PREC_FAST)
    best = min((ra.best_last, ra.best_ordinal), (rb.best_last, rb.best_ordinal))
    assert best[1] == res.best_ordinal


def test_config5_fleet_batch(solver):
    n_problems, n = 4096, 8192
    probs, seeds = W.fleet_problems(n_problems, depth=20)
    res = solver.eval_seeded_batch(probs, seeds, n, M.PREC_FAST)
    assert all(r.status == 0 for r in res)
    # a few vessels scored completely by the oracle
    for k in (0, 1, 777, 4095):
        o_last, o_cum, o_best = oracle_seeded(probs[k], seeds[k], 0, n)
        assert res[k].best_ordinal == o_best.best_index, k
        assert rel_err(res[k].best_last, o_best.best_last) <= TOL_FAST
    # every 64th vessel: the winner's oracle score beats a strided oracle sample of its own candidate set
    for k in range(0, n_problems, 64):
        w_last, _, _ = oracle_seeded(probs[k], seeds[k], int(res[k].best_ordinal), 1)
        s_last, _, _ = oracle_seeded(probs[k], seeds[k], 0, n // 128, stride=128)
        assert rel_err(res[k].best_last, w_last[0]) <= TOL_FAST and w_last[0] <= s_last.min() * (1 + 1e-12), k
    # a vessel solved alone = the same vessel inside the batch, bit for bit
    alone = solver.eval_seeded(probs[1234], seeds[1234], 0, n, M.PREC_FAST)
    assert alone.best_ordinal == res[1234].best_ordinal and alone.best_last == res[1234].best_last
    assert res[0].rollout_steps == n * 20 * 20
