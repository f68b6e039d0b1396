"""GPU parity tests proper: the HIP path (through the C-ABI) against the CPU oracle on identical seeded inputs.

Tolerances are the ones BASELINE.json's north_star states: per-candidate cost within 1e-5 relative in the
fast ("FP32") mode, within 1e-12 relative in the FP64 validation mode, and the SAME chosen candidate/control.
"""
import numpy as np
import pytest

import mpc_amd as M
from mpc_amd import workloads as W
from helpers import oracle_seeded, oracle_explicit, oracle_args, rel_err, traj_array
import oracle as O

pytestmark = pytest.mark.gpu

TOL_FAST = 1e-5
TOL_F64 = 1e-12


def _torch():
    import torch
    assert torch.cuda.is_available(), "these tests need the GPU"
    return torch


def _run_seeded(solver, p, seed, first, count, prec, selection=M.SELECT_LAST_STEP):
    torch = _torch()
    last = torch.empty(count, dtype=torch.float64, device="cuda")
    cum = torch.empty(count, dtype=torch.float64, device="cuda")
    res = solver.eval_seeded(p, seed, first, count, prec, selection, last, cum)
    return res, last.cpu().numpy(), cum.cpu().numpy()


@pytest.mark.parametrize("prec,tol", [(M.PREC_F64, TOL_F64), (M.PREC_FAST, TOL_FAST)])
@pytest.mark.parametrize("cfg", ["config2", "config3"])
def test_seeded_costs_and_choice(solver, prec, tol, cfg):
    p = W.config2_problem() if cfg == "config2" else W.config3_problem()
    n = 8192
    res, last, cum = _run_seeded(solver, p, W.SEED, 0, n, prec)
    o_last, o_cum, o_best = oracle_seeded(p, W.SEED, 0, n)
    e_last, e_cum = rel_err(last, o_last), rel_err(cum, o_cum)
    print("%s prec=%d max rel err last=%.3e cum=%.3e refined=%d kernel_ms=%.3f" % (
        cfg, prec, e_last.max(), e_cum.max(), res.refined, res.kernel_ms))
    assert e_last.max() <= tol and e_cum.max() <= tol
    assert res.status == 0
    assert res.best_ordinal == o_best.best_index
    assert rel_err(res.best_last, o_best.best_last) <= tol
    assert rel_err(res.best_cumulative, o_best.best_cumulative) <= tol
    assert res.evaluated == n and res.rollout_steps == n * p.depth * p.n_substeps


@pytest.mark.parametrize("prec", [M.PREC_F64, M.PREC_FAST])
def test_explicit_controls_match_seeded_and_oracle(solver, prec):
    torch = _torch()
    p = W.config2_problem()
    n = 4096
    rud, thr = W.seeded_controls(W.SEED, 0, n, p.depth, p.last_rudder, p.last_throttle)
    d_r = torch.tensor(rud, dtype=torch.float32, device="cuda").contiguous()
    d_t = torch.tensor(thr, dtype=torch.float32, device="cuda").contiguous()
    last = torch.empty(n, dtype=torch.float64, device="cuda")
    res = solver.eval_candidates(p, d_r, d_t, n, prec, M.SELECT_LAST_STEP, last, None)
    o_last, o_cum, o_best = oracle_explicit(p, rud, thr)
    assert rel_err(last.cpu().numpy(), o_last).max() <= (TOL_F64 if prec == M.PREC_F64 else TOL_FAST)
    assert res.best_ordinal == o_best.best_index
    res2, last2, _ = _run_seeded(solver, p, W.SEED, 0, n, prec)
    assert np.array_equal(last2, last.cpu().numpy())          # same kernel arithmetic, same bits
    assert res2.best_ordinal == res.best_ordinal


@pytest.mark.parametrize("prec,tol", [(M.PREC_F64, TOL_F64), (M.PREC_FAST, 1e-8)])
def test_winner_trajectory_matches_oracle_rollout(solver, prec, tol):
    p = W.config3_problem()
    res, _, _ = _run_seeded(solver, p, W.SEED, 0, 4096, prec)
    cfg, veh, cps, sub = oracle_args(p)
    rud, thr = W.seeded_controls(W.SEED, res.best_ordinal, 1, p.depth, p.last_rudder, p.last_throttle)
    traj = O.rollout(cfg, veh, cps, p.depth, sub, rud[:, 0], thr[:, 0])
    got = traj_array(res)
    for l in range(p.depth + 1):
        want = [traj[l].v.s.x, traj[l].v.s.y, traj[l].v.s.heading, traj[l].v.s.speed, traj[l].v.rpm]
        assert rel_err(got[l, :5], want).max() <= tol, l
        if l:
            assert rel_err(got[l, 5:7], [traj[l].v.cmg, traj[l].v.sog]).max() <= max(tol, 1e-10)
            assert got[l, 7] == traj[l].last_rudder and got[l, 8] == traj[l].last_throttle
            assert rel_err(got[l, 9], traj[l].score) <= max(tol, 1e-6 if prec == M.PREC_FAST else 0)
    assert res.first_rudder == rud[0, 0] and res.first_throttle == thr[0, 0]


def test_cumulative_selection(solver):
    p = W.config2_problem()
    n = 4096
    res, last, cum = _run_seeded(solver, p, W.SEED, 0, n, M.PREC_FAST, M.SELECT_CUMULATIVE)
    o_last, o_cum, o_best = oracle_seeded(p, W.SEED, 0, n, cumulative=True)
    assert res.best_ordinal == o_best.best_index
    assert rel_err(res.best_cumulative, o_best.best_cumulative) <= TOL_FAST


def test_near_ties_are_decided_by_the_strict_model(solver):
    """Duplicate control sequences give exactly equal scores: the lowest ordinal must win (controller.cpp:160
    is a strict <), in both precisions, and the fast mode must route the tie through the strict kernel."""
    torch = _torch()
    p = W.config2_problem(depth=8)
    n = 1024
    rud, thr = W.seeded_controls(5, 0, n, p.depth, p.last_rudder, p.last_throttle)
    o_last, _, o_best = oracle_explicit(p, rud, thr)
    b = int(o_best.best_index)
    for dup in (b + 7, b + 130, (b + 515) % n):      # copies of the best sequence at other ordinals
        rud[:, dup % n] = rud[:, b]
        thr[:, dup % n] = thr[:, b]
    o_last, _, o_best = oracle_explicit(p, rud, thr)
    d_r = torch.tensor(rud, dtype=torch.float32, device="cuda").contiguous()
    d_t = torch.tensor(thr, dtype=torch.float32, device="cuda").contiguous()
    for prec in (M.PREC_F64, M.PREC_FAST):
        res = solver.eval_candidates(p, d_r, d_t, n, prec)
        assert res.best_ordinal == o_best.best_index
        if prec == M.PREC_FAST:
            assert res.refined >= 2
            assert rel_err(res.best_last, o_best.best_last) <= TOL_F64    # refined winners are strict


def test_shard_offsets_and_ragged_sizes(solver):
    """first_ordinal != 0 (a GPU's shard of a larger set) and sizes that are not multiples of the wave size."""
    p = W.config2_problem(depth=6)
    for first, count in [(0, 1), (5, 63), (1000, 65), (123456789, 257)]:
        res, last, cum = _run_seeded(solver, p, 77, first, count, M.PREC_F64)
        o_last, o_cum, o_best = oracle_seeded(p, 77, first, count)
        assert rel_err(last, o_last).max() <= TOL_F64
        assert res.best_ordinal == o_best.best_index


def test_large_set_uses_grid_wide_compaction(solver):
    """> 2^18 candidates per problem: reduce/compact kernels build the refine list; checked on a strided sample
    plus the winner (size-independent property: the winner's oracle score is <= every sampled oracle score)."""
    p = W.config2_problem(depth=4)
    n = (1 << 18) + 4097
    res, last, cum = _run_seeded(solver, p, 99, 0, n, M.PREC_FAST)
    stride = 1021
    cnt = (n + stride - 1) // stride
    o_last, _, _ = oracle_seeded(p, 99, 0, cnt, stride=stride)
    assert rel_err(last[::stride], o_last).max() <= TOL_FAST
    w_last, _, _ = oracle_seeded(p, 99, int(res.best_ordinal), 1)
    assert rel_err(res.best_last, w_last[0]) <= TOL_FAST
    assert res.best_ordinal == int(np.argmin(last))            # first minimum of the device costs
    assert w_last[0] <= o_last.min() * (1 + 1e-12)


def test_batch_of_problems(solver):
    probs, seeds = W.fleet_problems(24, depth=8)
    n = 512
    res = solver.eval_seeded_batch(probs, seeds, n, M.PREC_FAST)
    for p, s, r in zip(probs, seeds, res):
        o_last, o_cum, o_best = oracle_seeded(p, s, 0, n)
        assert r.best_ordinal == o_best.best_index
        assert rel_err(r.best_last, o_best.best_last) <= TOL_FAST


def test_single_rollout_strict_and_fast(solver):
    """VehicleState::simulate itself through the C-ABI, incl. the reference-native 11-substep level."""
    p, _plan = W.config1_problem(depth=5, current=(0.3, -0.2))
    cfg, veh, cps, sub = oracle_args(p)
    rud = [0.5, -1.0, 0.25, 1.5, 0.0][:p.depth]
    thr = [0.75, 1.0, 0.0, -0.5, 0.3][:p.depth]
    traj = O.rollout(cfg, veh, cps, p.depth, sub, rud, thr)
    for prec, tol in [(M.PREC_F64, TOL_F64), (M.PREC_FAST, 1e-9)]:
        got = traj_array(solver.rollout(p, rud, thr, prec))
        for l in range(1, p.depth + 1):
            want = [traj[l].v.s.x, traj[l].v.s.y, traj[l].v.s.heading, traj[l].v.s.speed, traj[l].v.rpm]
            assert rel_err(got[l, :5], want).max() <= tol, (prec, l, got[l, :5], want)
            assert rel_err(got[l, 9], traj[l].score) <= max(tol, 1e-6 if prec == M.PREC_FAST else 0), (prec, l)


def test_stationary_vessel_edge_case(solver):
    """speed 0, throttle 0, no current: dx = dy = 0 in every substep, atan2(0, 0) = 0 (VehicleState.h:143)."""
    torch = _torch()
    p = W.config2_problem(depth=3)
    p.start_speed = 0.0
    n = 64
    rud = np.zeros((3, n)); thr = np.zeros((3, n))
    rud[:, 1:] = np.linspace(-1, 1, n - 1).astype(np.float32).astype(np.float64)[None, :]   # the ABI takes FP32 controls
    thr[:, 32:] = 0.5
    o_last, o_cum, o_best = oracle_explicit(p, rud, thr)
    d_r = torch.tensor(rud, dtype=torch.float32, device="cuda").contiguous()
    d_t = torch.tensor(thr, dtype=torch.float32, device="cuda").contiguous()
    for prec, tol in [(M.PREC_F64, TOL_F64), (M.PREC_FAST, TOL_FAST)]:
        last = torch.empty(n, dtype=torch.float64, device="cuda")
        res = solver.eval_candidates(p, d_r, d_t, n, prec, M.SELECT_LAST_STEP, last, None)
        assert rel_err(last.cpu().numpy(), o_last).max() <= tol
        assert res.best_ordinal == o_best.best_index


def test_errors_are_reported_not_swallowed(solver):
    p = W.config2_problem()
    with pytest.raises(M.MpcError):
        solver.eval_seeded(p, 1, 0, 0)                       # zero candidates
    q = W.config2_problem()
    q.substeps[0] = 0.25                                     # fast mode needs substeps <= 0.1 s
    with pytest.raises(M.MpcError):
        solver.eval_seeded(q, 1, 0, 64, M.PREC_FAST)
    q.depth = 0
    with pytest.raises(M.MpcError):
        solver.eval_seeded(q, 1, 0, 64, M.PREC_F64)


def test_combine_shards_kernel(solver):
    """mpc_hip_combine_shards against the torch form the gloo tests use, on synthetic gathered winners."""
    torch = _torch()
    from mpc_amd.distributed import combine_gathered
    rng = np.random.RandomState(5)
    shards, P = 8, 777
    scores = rng.choice([0.5, 1.0, 2.0, 3.5, np.inf], size=(shards, P))
    ords = rng.randint(0, 65536, size=(shards, P)).astype(np.int64)
    ords[scores == np.inf] = -1
    g = np.stack([scores.view(np.int64), ords], axis=-1)
    d_g = torch.tensor(g, dtype=torch.int64, device="cuda").contiguous()
    d_out = torch.zeros(2 * P, dtype=torch.int64, device="cuda")
    solver.combine_shards(d_g, shards, P, 65536, d_out)
    solver.synchronize()
    key, cand = combine_gathered(d_g.view(-1), shards, P, 65536)
    got = d_out.view(-1, 2)
    assert torch.equal(got[:, 0], key)
    want = torch.where(cand == np.iinfo(np.int64).max, torch.full_like(cand, -1), cand)
    assert torch.equal(got[:, 1], want)


def test_shard_combiner_single_rank(solver):
    """ShardCombiner (the batched, pipelined combine bench.py uses) with one rank: the combined winner of every
    solve is that solve's own winner; partial batches and ring wrap-around included."""
    from mpc_amd.distributed import ShardCombiner, bits_to_score
    p = W.config2_problem(depth=5)
    for use_sink in (True, False):           # winners deposited by the finish kernel itself / copied afterwards
        comb = ShardCombiner(solver, 1, ordinal_stride=0, batch=3, slots=2)
        want, seen = None, []
        for k in range(8):                   # 2 full batches + a partial one
            solver.upload([p], [100 + k])
            if use_sink:
                comb.begin()
            solver.run_seeded(0, 2048, M.PREC_FAST)
            comb.submit()
            want = solver.fetch()[0]
            seen.append(want.best_ordinal)
        key, ordinal = comb.flush()
        assert int(ordinal[0].item()) == want.best_ordinal
        assert bits_to_score(key)[0] == want.best_last
        assert len(set(seen)) > 1            # different seeds, different winners: rows are not mixed up
        # the last batch holds solves 6 and 7
        last_batch = comb.results.view(comb.B, 1, 2)[:2, 0, 1].cpu().tolist()
        assert last_batch == seen[6:8]


def test_timing_period_samples_runs_and_changes_no_result(solver):
    """mpc_hip_set_timing: events ride on every n-th run only; results do not depend on it."""
    p = W.config2_problem()
    solver.upload([p], [W.SEED])
    try:
        solver.set_timing(1)
        solver.run_seeded(0, 4096, M.PREC_FAST)
        ref = solver.fetch()[0]
        assert ref.kernel_ms > 0 and ref.total_ms >= ref.kernel_ms
        solver.timing(reset=True)
        solver.set_timing(3)
        for _ in range(9):
            solver.run_seeded(0, 4096, M.PREC_FAST)
        runs, k_ms, t_ms = solver.timing(reset=True)
        assert runs == 3 and 0 < k_ms <= t_ms
        solver.set_timing(0)
        solver.run_seeded(0, 4096, M.PREC_FAST)
        res = solver.fetch()[0]
        assert res.kernel_ms == 0 and res.total_ms == 0
        assert (res.best_ordinal, res.best_last, res.first_rudder) == (ref.best_ordinal, ref.best_last, ref.first_rudder)
        assert solver.timing(reset=True)[0] == 0
    finally:
        solver.set_timing(1)


def test_shard_combiner_exact_pipeline_one_rank_rccl():
    """The pipeline bench.py --gpus N runs (winner records with controls -> one all-gather -> exact combine),
    rehearsed over RCCL with a one-rank group on this GPU: every solve's combined winner is its own winner,
    through full batches, a partial one and slot wrap-around; a lone shard never ties."""
    import os
    import torch
    import torch.distributed as dist
    from mpc_amd.distributed import ShardCombiner, bits_to_score
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29541")
    created = not dist.is_initialized()
    if created:
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    stream = torch.cuda.Stream()
    try:
        with torch.cuda.stream(stream):
            solver = M.Solver(0, stream=stream.cuda_stream)          # collectives and kernels ordered on one stream
            p = W.config2_problem(depth=5)
            comb = ShardCombiner(solver, 1, ordinal_stride=0, batch=3, slots=3, always=True, exact=True, depth=5)
            assert comb.exact and comb.collective
            seen = []
            for k in range(11):                                      # 3 full batches + a partial one: slots wrap
                solver.upload([p], [200 + k])
                comb.begin()
                solver.run_seeded(0, 2048, M.PREC_FAST)
                comb.submit()
                seen.append(solver.fetch()[0])
            key, ordinal = comb.flush()
            assert int(ordinal[0].item()) == seen[-1].best_ordinal and bits_to_score(key)[0] == seen[-1].best_last
            assert comb.results.view(comb.B, 1, 2)[:2, 0, 1].cpu().tolist() == [r.best_ordinal for r in seen[9:11]]
            assert comb.tied_rows() == 0
            solver.close()
    finally:
        if created:
            dist.destroy_process_group()
