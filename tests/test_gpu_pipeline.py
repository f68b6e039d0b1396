"""mpc_hip_set_pipeline: a stream of solves whose finish kernels ride in the NEXT solve's rollout launch returns, bit
for bit, what the two-kernel form returns -- winners deposited through the sink, results fetched at the end or in
between, runs that cannot fuse (strict precision, reference-native schedule, another candidate source) mixed in."""
import numpy as np
import pytest

import mpc_amd as M
from mpc_amd import workloads as W
from helpers import traj_array

pytestmark = pytest.mark.gpu


def _torch():
    import torch
    assert torch.cuda.is_available()
    return torch


def _stream_of_solves(solver, pipeline, plan):
    """plan: list of (kind, arg...) steps; returns every observable, in order"""
    torch = _torch()
    solver.set_pipeline(pipeline)
    seen = []
    p = W.config2_problem(depth=8)
    p11 = M.make_problem((0.5, -0.5, 1.5, 1.8), W.config2_plan().checkpoints(W.T0, 6), M.substep_schedule(1.0),
                         last=(0.1, 0.6), current=(0.2, -0.1))
    n = 4096
    rud, thr = W.seeded_controls(W.SEED + 5, 0, n, 8, p.last_rudder, p.last_throttle)
    d_r = torch.tensor(rud.astype(np.float32), device="cuda").contiguous()
    d_t = torch.tensor(thr.astype(np.float32), device="cuda").contiguous()
    sinks = torch.full((len(plan), 2), -7, dtype=torch.int64, device="cuda")
    torch.cuda.synchronize()
    solver.upload([p], [W.SEED])
    for k, step in enumerate(plan):
        solver.set_winner_sink(sinks[k])
        if step == "seeded":
            solver.run_seeded(k * 1000, n, M.PREC_FAST)
        elif step == "seeded_small":
            solver.run_seeded(k * 1000, 777, M.PREC_FAST)
        elif step == "strict":
            solver.run_seeded(k * 1000, 2048, M.PREC_F64)
        elif step == "explicit":
            solver.run_candidates(d_r, d_t, n, M.PREC_FAST)
        elif step == "cumulative":
            solver.run_seeded(k * 1000, n, M.PREC_FAST, M.SELECT_CUMULATIVE)
        elif step == "native":
            solver.upload([p11], [W.SEED + k])
            solver.run_seeded(0, n, M.PREC_FAST)
        elif step == "reupload":
            solver.upload([p], [W.SEED + k])
            solver.run_seeded(0, n, M.PREC_FAST)
        elif step == "static":                                # the entry point that allocates nothing never defers
            solver.run_seeded_static(k * 1000, 2048, M.PREC_FAST)
        elif step == "fetch":
            r = solver.fetch()[0]
            seen.append(("fetch", r.best_ordinal, r.best_last, r.best_cumulative, traj_array(r).tobytes()))
            solver.run_seeded(k * 1000, n, M.PREC_FAST)
        else:
            raise AssertionError(step)
    solver.set_winner_sink(None)
    r = solver.fetch()[0]
    seen.append(("last", r.best_ordinal, r.best_last, r.best_cumulative, traj_array(r).tobytes()))
    solver.synchronize()
    seen.append(("sinks", sinks.cpu().numpy().tobytes()))
    solver.set_pipeline(False)
    return seen


@pytest.mark.parametrize("plan", [
    ["seeded"] * 6,
    ["seeded"] * 7,
    ["explicit"] * 5,
    ["explicit"] * 8,
    ["seeded", "seeded", "explicit", "explicit", "explicit", "seeded", "seeded_small", "seeded"],
    ["seeded", "seeded_small", "seeded", "cumulative", "seeded"],
    ["seeded", "strict", "seeded", "explicit", "seeded", "native", "seeded", "reupload", "seeded"],
    ["seeded", "fetch", "seeded", "fetch", "explicit", "explicit", "fetch"],
])
def test_deferred_finish_changes_no_result(plan):
    solver = M.Solver(0)
    try:
        want = _stream_of_solves(solver, False, plan)
        for level in (1, 2, 3, 4):                            # 1: deferred finish; K >= 2: K runs per launch as well
            got = _stream_of_solves(solver, level, plan)
            assert len(got) == len(want)
            for a, b in zip(got, want):
                assert a == b, (level, a[0])
        sinks = np.frombuffer(want[-1][1], dtype=np.int64).reshape(-1, 2)
        assert (sinks[:, 1] >= 0).all()                       # every run deposited its winners in ITS row
    finally:
        solver.close()


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_random_streams_give_the_same_results_at_every_pipeline_level(seed):
    """40 random steps (every kind of run, fetches in between) at levels 0, 1 and 2: every sink row, every fetched
    result and the final one are bit for bit the same -- whatever pairs up, rides along or goes alone"""
    rng = np.random.default_rng(seed)
    kinds = ["seeded", "seeded", "explicit", "explicit", "seeded_small", "cumulative", "strict", "fetch", "reupload", "native", "static"]
    plan = [kinds[int(k)] for k in rng.integers(0, len(kinds), 40)]
    solver = M.Solver(0)
    try:
        want = _stream_of_solves(solver, 0, plan)
        for level in (1, 2, 3, 4):
            got = _stream_of_solves(solver, level, plan)
            assert len(got) == len(want)
            for a, b in zip(got, want):
                assert a == b, (level, a[0], plan)
    finally:
        solver.close()


def test_deferred_finish_on_a_fleet_and_with_timing():
    solver = M.Solver(0)
    try:
        probs, seeds = W.fleet_problems(24, depth=10)
        out = []
        for pipeline in (0, 1, 2, 3, 4):
            solver.set_pipeline(pipeline)
            solver.set_timing(1)
            solver.upload(probs, seeds)
            solver.timing(reset=True)
            for k in range(4):
                solver.run_seeded(0, 1024 + 64 * k, M.PREC_FAST)
            runs, k_ms, t_ms = solver.timing(reset=True)
            # level K >= 2: one pair of events per launch of K runs (4 runs: two launches at K = 2 and K = 3, one at K = 4)
            assert runs == {0: 4, 1: 4, 2: 2, 3: 2, 4: 1}[pipeline] and 0 < k_ms <= t_ms
            res = solver.fetch()
            out.append([(r.status, r.best_ordinal, r.best_last, traj_array(r).tobytes()) for r in res])
        assert out[0] == out[1] == out[2] == out[3] == out[4]
        solver.set_timing(1)
    finally:
        solver.close()


@pytest.mark.parametrize("pipeline", [0, 1, 2, 3, 4])
def test_winner_record_sink_carries_the_winners_controls(pipeline):
    """mpc_hip_set_winner_record_sink / mpc_hip_copy_winner_records: (score bits, ordinal) as the plain sink writes
    them, then the winner's clamped controls per level -- the ones in the fetched trajectory, bit for bit -- for every
    size of candidate set (one lane, snapshot finish, two-kernel finish), explicit and seeded, on a fleet whose
    problems are shallower than the record, immediate and deferred finish."""
    torch = _torch()
    solver = M.Solver(0)
    try:
        depth, rec_depth = 9, 12
        Wd = M.winner_record_words(rec_depth)
        probs, seeds = W.fleet_problems(5, depth=depth)
        solver.upload(probs, seeds)
        solver.set_pipeline(pipeline)
        sizes = [1, 60, 777, 4096, 70000]
        sink = torch.full((len(sizes), 5, Wd), -7, dtype=torch.int64, device="cuda")
        torch.cuda.synchronize()
        fetched = []
        for k, n in enumerate(sizes):
            solver.set_winner_record_sink(sink[k], rec_depth)
            solver.run_seeded(100 * k, n, M.PREC_FAST)
            packed = torch.full((5, Wd), -9, dtype=torch.int64, device="cuda")
            torch.cuda.synchronize()
            solver.copy_winner_records(packed, rec_depth)            # (flushes a deferred finish)
            fetched.append((solver.fetch(), packed))
        solver.set_winner_record_sink(None, rec_depth)
        solver.synchronize()
        torch.cuda.synchronize()
        for k, (res, packed) in enumerate(fetched):
            for q in range(5):
                rec = sink[k, q].cpu().numpy()
                assert rec[0] == np.float64(res[q].best_last).view(np.int64) and rec[1] == res[q].best_ordinal
                tr = traj_array(res[q])                              # [depth + 1, 7]: x y heading speed time rudder throttle? see helpers
                ctl = rec[2:2 + 2 * depth].view(np.float64).reshape(depth, 2)
                want = np.array([[res[q].trajectory[l + 1].rudder, res[q].trajectory[l + 1].throttle] for l in range(depth)])
                assert np.array_equal(ctl.view(np.int64), want.view(np.int64)), (k, q)
                assert (rec[2 + 2 * depth:] == -7).all()             # nothing beyond the problem's depth is touched
                assert np.array_equal(packed[q].cpu().numpy()[:2 + 2 * depth], rec[:2 + 2 * depth])
        with pytest.raises(M.MpcError):                              # a record shallower than a resident problem
            solver.set_winner_record_sink(sink[0], depth - 1)
            solver.run_seeded(0, 64, M.PREC_FAST)
        solver.set_winner_record_sink(None, rec_depth)
        solver.set_pipeline(False)
    finally:
        solver.close()


@pytest.mark.parametrize("pipeline", [0, 1, 2, 3, 4])
def test_a_control_buffer_rewritten_between_runs_needs_a_flush_first(pipeline):
    """include/mpc_hip.h, the deferred finish: a run's explicit control arrays are still read by its PENDING finish (the
    strict decision among near-ties, the winner's trajectory and record), so a sampling loop that rewrites ONE control
    buffer in place calls mpc_hip_flush before each rewrite.  Done that way, every run's record and result are what the
    same set gives when solved alone -- at every pipeline level."""
    torch = _torch()
    p = W.config2_problem(depth=8)
    n, rounds = 4096, 5
    host = [W.seeded_controls(W.SEED + 40 + k, 0, n, 8, p.last_rudder, p.last_throttle) for k in range(rounds)]
    alone = M.Solver(0)
    want = []
    for rud, thr in host:
        r = alone.eval_candidates(p, torch.tensor(rud, dtype=torch.float32, device="cuda").contiguous(),
                                  torch.tensor(thr, dtype=torch.float32, device="cuda").contiguous(), n, M.PREC_FAST)
        want.append((int(r.best_ordinal), r.best_last, [r.trajectory[l + 1].rudder for l in range(8)], traj_array(r).tobytes()))
    alone.close()
    stream = torch.cuda.Stream()
    with torch.cuda.stream(stream):                               # the rewrites (torch copies) run on the solver's stream
        solver = M.Solver(0, stream=stream.cuda_stream)
        solver.upload([p], [W.SEED])
        solver.set_pipeline(pipeline)
        d_r = torch.empty((8, n), dtype=torch.float32, device="cuda")
        d_t = torch.empty((8, n), dtype=torch.float32, device="cuda")
        W8 = M.winner_record_words(8)
        recs = torch.zeros((rounds, W8), dtype=torch.int64, device="cuda")
        last = None
        for k, (rud, thr) in enumerate(host):
            solver.flush()                                        # BEFORE the buffer changes: the pending finish reads it
            d_r.copy_(torch.tensor(rud, dtype=torch.float32), non_blocking=False)
            d_t.copy_(torch.tensor(thr, dtype=torch.float32), non_blocking=False)
            solver.set_winner_record_sink(recs[k], 8)
            solver.run_candidates(d_r, d_t, n, M.PREC_FAST)
        last = solver.fetch()[0]
        solver.set_winner_record_sink(None, 8)
        torch.cuda.synchronize()
        got = recs.cpu().numpy()
        for k in range(rounds):
            assert int(got[k, 1]) == want[k][0], (pipeline, k)
            assert got[k, 0] == np.float64(want[k][1]).view(np.int64)
            assert [float(v) for v in got[k, 2::2].view(np.float64)] == want[k][2]
        assert traj_array(last).tobytes() == want[-1][3]
        solver.close()
