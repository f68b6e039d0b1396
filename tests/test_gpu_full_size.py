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


def test_config4_the_whole_16M_set_as_8_shards_and_as_one_launch():
    """BASELINE config 4 at its real size: 2^24 device-generated candidates x 30 s x 20 substeps.  The 8 contiguous
    shards a node's 8 GPUs would hold run one after the other on this GPU (one handle each, as one process per GPU
    holds one), their packed winners are gathered and combined (a) by mpc_hip_combine_shards on the fast keys and (b)
    by the exact decision (winner records with controls -> mpc_hip_combine_shards_exact); both must be the winner of ONE 2^24-candidate
    launch, whose costs the shards reproduce bit for bit, and a strided 1/4096 oracle sample over the whole ordinal
    range is within 1e-5 and never better than the winner (controller.cpp:158-166: first strict minimum)."""
    torch = _torch()
    from mpc_amd.distributed import combine_shards_exact
    p = W.config2_problem(depth=30)
    total, world = 1 << 24, 8
    whole = M.Solver(0)
    shards = [M.Solver(0) for _ in range(world)]
    try:
        last_all = torch.empty(total, dtype=torch.float64, device="cuda")
        res_all = whole.eval_seeded(p, W.SEED, 0, total, M.PREC_FAST, M.SELECT_LAST_STEP, last_all, None)
        assert res_all.status == 0 and res_all.evaluated == total and res_all.rollout_steps == total * 30 * 20
        last_sh = torch.empty(total, dtype=torch.float64, device="cuda")
        wins = torch.zeros((world, 1, 2), dtype=torch.int64, device="cuda")
        recs = torch.zeros((world, 1, M.winner_record_words(30)), dtype=torch.int64, device="cuda")
        torch.cuda.synchronize()
        for s, solver in enumerate(shards):
            first, count = shard_range(total, s, world)
            assert count == 1 << 21
            solver.upload([p], [W.SEED])
            solver.run_seeded(first, count, M.PREC_FAST, M.SELECT_LAST_STEP, last_sh[first:first + count], None)
            solver.copy_winners(wins[s])
            solver.copy_winner_records(recs[s], 30)
            r = solver.fetch()[0]
            assert r.status == 0 and first <= r.best_ordinal < first + count
        assert torch.equal(last_sh.view(torch.int64), last_all.view(torch.int64))         # every candidate, bit for bit
        out_fast = torch.zeros((1, 2), dtype=torch.int64, device="cuda")
        torch.cuda.synchronize()
        shards[0].combine_shards(wins, world, 1, 0, out_fast)
        shards[0].synchronize()
        assert torch.equal(recs[:, :, :2], wins)
        out_exact = combine_shards_exact(shards[3], recs, 30, 0)
        best = int(res_all.best_ordinal)
        assert int(out_fast[0, 1]) == best and int(out_exact[0, 1]) == best
        key = int(np.float64(res_all.best_last).view(np.int64))
        assert int(out_fast[0, 0]) == key and int(out_exact[0, 0]) == key
        # the device argmin is the first minimum of the device costs
        host = last_all.cpu().numpy()
        assert best == int(np.argmin(host)) and host[best] == res_all.best_last
        # oracle: strided sample of the WHOLE range + the winner itself
        stride = 4096
        o_last, _, _ = oracle_seeded(p, W.SEED, 0, total // stride, stride=stride)
        assert rel_err(host[::stride], o_last).max() <= TOL_FAST
        w_last, _, _ = oracle_seeded(p, W.SEED, best, 1)
        assert rel_err(res_all.best_last, w_last[0]) <= TOL_FAST and w_last[0] <= o_last.min()
    finally:
        whole.close()
        for s in shards:
            s.close()


def test_cross_shard_near_tie_is_decided_by_the_strict_model():
    """Two shards whose winners lie within MPC_REFINE_REL of each other and whose FAST scores order them the other way
    round than the reference's arithmetic does: the fast-key combine picks the wrong shard, the exact decision the
    right one.  The pair is searched among 2^20 candidates of config 2 (adjacent in strict order, inverted in fast
    order), then each becomes a one-candidate shard, in both shard orders."""
    torch = _torch()
    from mpc_amd.distributed import combine_shards_exact
    p = W.config2_problem(depth=20)
    n = 1 << 20
    s0, s1 = M.Solver(0), M.Solver(0)
    try:
        d = torch.empty(n, dtype=torch.float64, device="cuda")
        s0.eval_seeded(p, W.SEED, 0, n, M.PREC_FAST, M.SELECT_LAST_STEP, d, None)
        fast = d.cpu().numpy().copy()
        s0.eval_seeded(p, W.SEED, 0, n, M.PREC_F64, M.SELECT_LAST_STEP, d, None)
        strict = d.cpu().numpy().copy()
        order = np.argsort(strict, kind="stable")
        fs, ss = fast[order], strict[order]
        flips = np.flatnonzero((ss[1:] > ss[:-1]) & (fs[1:] < fs[:-1]) & np.isfinite(ss[1:]))
        assert flips.size > 0, "no pair of candidates whose fast scores invert their strict order"
        a, b = int(order[flips[0]]), int(order[flips[0] + 1])              # strict: a < b; fast: b < a
        assert strict[a] < strict[b] and fast[b] < fast[a] and fast[a] <= fast[b] * (1 + M.REFINE_REL)
        ctl = {}
        for o in (a, b):
            r, t = W.seeded_controls(W.SEED, o, 1, 20, p.last_rudder, p.last_throttle)
            ctl[o] = (torch.tensor(r.astype(np.float32), device="cuda").contiguous(),
                      torch.tensor(t.astype(np.float32), device="cuda").contiguous())
        for first_shard, second_shard in ((b, a), (a, b)):
            wins = torch.zeros((2, 1, 2), dtype=torch.int64, device="cuda")
            recs = torch.zeros((2, 1, M.winner_record_words(20)), dtype=torch.int64, device="cuda")
            torch.cuda.synchronize()
            for s, (solver, o) in enumerate(((s0, first_shard), (s1, second_shard))):
                res = solver.eval_candidates(p, ctl[o][0], ctl[o][1], 1, M.PREC_FAST)
                assert res.best_ordinal == 0 and res.best_last == fast[o]                # one candidate: local ordinal 0
                solver.copy_winners(wins[s])
                solver.copy_winner_records(recs[s], 20)
                solver.synchronize()
            out_fast = torch.zeros((1, 2), dtype=torch.int64, device="cuda")
            torch.cuda.synchronize()
            s0.combine_shards(wins, 2, 1, 1, out_fast)                                   # ordinal stride 1: global = shard index
            s0.synchronize()
            out_exact = combine_shards_exact(s0, recs, 20, 1)
            shard_of = {first_shard: 0, second_shard: 1}
            assert int(out_fast[0, 1]) == shard_of[b]                                    # the fast keys prefer b ...
            assert int(out_exact[0, 1]) == shard_of[a]                                   # ... the reference's arithmetic a
            assert int(out_exact[0, 0]) == int(np.float64(strict[a]).view(np.int64))
        assert s0.tied_rows() == 2                                                       # both rounds had a near-tie row
    finally:
        s0.close()
        s1.close()
