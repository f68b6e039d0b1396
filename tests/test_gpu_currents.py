"""Device-side current estimation for fleets (mpc_hip_estimate_currents) against the oracle's restatement
of CurrentEstimator::getCurrent (CurrentEstimator.cpp:15-41), and a fleet solved with device-applied currents against
the same fleet uploaded with those currents from the host."""
import numpy as np
import pytest

import mpc_amd as M
from mpc_amd import workloads as W
import oracle as O

pytestmark = pytest.mark.gpu

TOL = 1e-12          # FP64 validation bar: ocml vs libm sin/cos over <= 1200 substeps of a ~100-m track


def _history(rng, n_samples, current, hz=10.0, jitter=0.02, start=(5.0, -3.0, 1.0, 2.0, 100.0)):
    """a vessel driven by random controls under a true current, sampled every ~1/hz s -> ([HistoryEntry], final Vehicle)"""
    v = O.vehicle(*start)
    hist = []
    r, t = 0.0, 0.5
    for i in range(n_samples):
        if i % 7 == 0:
            r, t = float(rng.randint(-16, 17)) * 0.0625, float(rng.randint(0, 9)) * 0.125
        hist.append(O.HistoryEntry(v.s, r, t))
        v = O.simulate(v, r, t, 1.0 / hz + rng.uniform(-jitter, jitter), current[0], current[1])
    return hist, v


def _as_samples(hist):
    return [(e.s.x, e.s.y, e.s.heading, e.s.speed, e.s.time, e.rudder, e.throttle) for e in hist]


def _check(solver, cases):
    """cases: [(history, actual State)]"""
    got = solver.estimate_currents([_as_samples(h) for h, _ in cases], [(a.x, a.y, a.time) for _, a in cases])
    for k, ((h, a), g) in enumerate(zip(cases, got)):
        want = O.current_estimate(h, True, a)
        assert abs(g[0] - want[0]) <= TOL * max(1.0, abs(want[0])) and abs(g[1] - want[1]) <= TOL * max(1.0, abs(want[1])), (k, g, want)
    return got


def test_ragged_fleet_of_histories(solver):
    """histories of 0 ... 600 samples in one call: below 10 samples the reference answers (0, 0)"""
    rng = np.random.RandomState(11)
    cases = []
    for n in (0, 1, 9, 10, 11, 30, 64, 200, 599, 600):
        for _ in range(3):
            cur = (rng.uniform(-0.5, 0.5), rng.uniform(-0.5, 0.5))
            start = (rng.uniform(-50, 50), rng.uniform(-50, 50), rng.uniform(0, 6.28), rng.uniform(0.0, 2.75), rng.uniform(0, 1000))
            hist, v = _history(rng, n, cur, start=start)
            cases.append((hist, v.s, cur, n))
    got = _check(solver, [(h, a) for h, a, _, _ in cases])
    for (h, a, cur, n), g in zip(cases, got):
        if n < M.binding.CURRENT_MIN_SAMPLES:
            assert g == (0.0, 0.0)
        elif n >= 200:                        # roughly the true current (the replay applies sample i's controls over the
            assert abs(g[0] - cur[0]) < 0.15 and abs(g[1] - cur[1]) < 0.15, (n, g, cur)      # interval BEFORE sample i)


def test_the_replays_branches(solver):
    rng = np.random.RandomState(5)
    hist, v = _history(rng, 40, (0.3, -0.15))
    inside = O.State(hist[25].s.x + 0.1, hist[25].s.y - 0.2, 0.0, 0.0, 0.5 * (hist[25].s.time + hist[26].s.time))
    later = O.simulate(v, 0.1, 0.6, 3.7, 0.3, -0.15).s                     # 3.7 s past the newest sample: hold its control
    exact = O.State(1.0, 2.0, 0.0, 0.0, hist[12].s.time)                   # query AT a sample time
    twin = list(hist); twin.insert(20, hist[20])                           # two samples with one time stamp: a 0-s simulate
    shuffled = list(hist); shuffled[10], shuffled[30] = shuffled[30], shuffled[10]     # non-monotonic times
    cases = [(hist, v.s), (hist, inside), (hist, later), (hist, exact), (twin, v.s), (shuffled, v.s),
             (hist, hist[0].s),                                            # query not after the oldest sample
             (hist, O.State(0.0, 0.0, 0.0, 0.0, hist[0].s.time - 5.0))]
    got = _check(solver, cases)
    assert got[6] == (0.0, 0.0) and got[7] == (0.0, 0.0)


def test_refused_inputs(solver):
    rng = np.random.RandomState(1)
    hist, v = _history(rng, 12, (0.0, 0.0))
    s = _as_samples(hist)
    with pytest.raises(M.MpcError) as e:
        solver.estimate_currents([s * 60], [(0.0, 0.0, v.s.time)])                      # 720 samples > 600
    assert e.value.status == 8
    with pytest.raises(M.MpcError) as e:
        solver.estimate_currents([s], [(0.0, 0.0, hist[0].s.time + 1.0e6)])             # a replay of 1e6 s
    assert e.value.status == 8
    with pytest.raises(M.MpcError):
        solver.estimate_currents([s], [(0.0, 0.0, float("nan"))])
    solver.upload([W.config2_problem(3)], [W.SEED])
    with pytest.raises(M.MpcError):
        solver.estimate_currents([s, s], [(0.0, 0.0, v.s.time)] * 2, apply=True)        # two histories, one problem


def test_fleet_with_applied_currents_equals_host_upload(solver):
    """plans -> checkpoints -> currents -> solve, all on the device, equals the fleet uploaded with the same currents"""
    n_problems, n, depth = 64, 2048, 20
    rng = np.random.RandomState(3)
    probs, seeds = W.fleet_problems(n_problems, depth)
    hists, actual = [], []
    for p in probs:
        hist, v = _history(rng, int(rng.randint(20, 120)), (p.current_x, p.current_y),
                           start=(p.start_x - 20.0, p.start_y + 10.0, p.start_heading, p.start_speed, 50.0))
        hists.append(_as_samples(hist)); actual.append((v.s.x, v.s.y, v.s.time))
    blank = []
    for p in probs:                                   # the device must not depend on the host's currents
        q = M.Problem.from_buffer_copy(p)
        q.current_x = q.current_y = 123.0
        blank.append(q)
    solver.upload(blank, seeds)
    est = solver.estimate_currents(hists, actual, apply=True)
    with pytest.raises(M.MpcError):
        solver.fetch()                                # no run yet with the new currents
    ref_probs = []
    for p, (cx, cy) in zip(probs, est):
        q = M.Problem.from_buffer_copy(p)
        q.current_x, q.current_y = cx, cy
        ref_probs.append(q)
    for prec in (M.PREC_F64, M.PREC_FAST):
        solver.upload(blank, seeds)
        solver.estimate_currents(hists, actual, apply=True)
        solver.run_seeded(0, n, prec)
        got = [(r.status, r.best_ordinal, r.best_last, r.best_cumulative, r.first_rudder, r.first_throttle) for r in solver.fetch()]
        want = [(r.status, r.best_ordinal, r.best_last, r.best_cumulative, r.first_rudder, r.first_throttle)
                for r in solver.eval_seeded_batch(ref_probs, seeds, n, prec)]
        assert got == want                            # the same records, bit for bit
        assert all(g[0] == 0 for g in got)


def test_device_resident_histories_equal_the_uploaded_ones():
    """mpc_hip_history_reset / _update / mpc_hip_estimate_currents_resident: CurrentEstimator's own state for a fleet kept
    in HBM.  A fleet of 37 vessels is ticked 700 times (the reference drops the first 20 updates after a reset and keeps
    the newest 600: CurrentEstimator.cpp:5-13); at several ticks the resident estimate must equal, bit for bit, the
    uploaded-history entry point given the history a CurrentEstimator would hold then, which in turn is checked against
    the oracle."""
    solver = M.Solver(0)
    try:
        rng = np.random.RandomState(23)
        n = 37
        vessels, kept = [], [[] for _ in range(n)]
        for p in range(n):
            cur = (rng.uniform(-0.4, 0.4), rng.uniform(-0.4, 0.4))
            start = (rng.uniform(-50, 50), rng.uniform(-50, 50), rng.uniform(0, 6.28), rng.uniform(0.5, 2.75), 100.0 + p)
            vessels.append([O.vehicle(*start), cur, 0.0, 0.5])
        solver.history_reset(n)
        skips = 20
        for tick in range(700):
            samples = []
            for p, ves in enumerate(vessels):
                v, cur, r, t = ves
                if (tick + p) % 9 == 0:
                    ves[2], ves[3] = r, t = float(rng.randint(-16, 17)) * 0.0625, float(rng.randint(0, 9)) * 0.125
                samples.append((v.s.x, v.s.y, v.s.heading, v.s.speed, v.s.time, r, t))
                if tick >= skips:
                    kept[p].append(O.HistoryEntry(v.s, r, t))
                    if len(kept[p]) > M.binding.CURRENT_MAX_SAMPLES:
                        kept[p].pop(0)
                ves[0] = O.simulate(v, r, t, 0.1 + rng.uniform(-0.01, 0.01), cur[0], cur[1])
            solver.history_update(samples)
            if tick in (5, 19, 25, 29, 30, 31, 200, 619, 620, 621, 699):
                actual = [(ves[0].s.x, ves[0].s.y, ves[0].s.time) for ves in vessels]
                resident = solver.estimate_currents_resident(actual)
                uploaded = solver.estimate_currents([_as_samples(h) for h in kept], actual)
                assert resident == uploaded, tick                                    # same arithmetic, same bits
                for p in (0, 17, 36):
                    want = O.current_estimate(kept[p], True, vessels[p][0].s)
                    assert abs(resident[p][0] - want[0]) <= TOL * max(1.0, abs(want[0])), (tick, p)
                    assert abs(resident[p][1] - want[1]) <= TOL * max(1.0, abs(want[1])), (tick, p)
                if tick < 29:
                    assert all(c == (0.0, 0.0) for c in resident)                    # 20 skipped + fewer than 10 kept
        assert len(kept[0]) == 600
        # apply: the estimates become the resident problems' currents, as the uploaded form does
        probs, seeds = W.fleet_problems(n, depth=6)
        solver.upload(probs, seeds)
        actual = [(ves[0].s.x, ves[0].s.y, ves[0].s.time) for ves in vessels]
        cur = solver.estimate_currents_resident(actual, apply=True)
        a = solver.eval_seeded_batch([_with_current(q, c) for q, c in zip(probs, cur)], seeds, 512, M.PREC_FAST)
        solver.upload(probs, seeds)
        solver.estimate_currents_resident(actual, apply=True)
        solver.run_seeded(0, 512, M.PREC_FAST)
        b = solver.fetch()
        assert [(r.best_ordinal, r.best_last) for r in a] == [(r.best_ordinal, r.best_last) for r in b]
        # a reset empties the rings
        solver.history_reset(n)
        assert all(c == (0.0, 0.0) for c in solver.estimate_currents_resident(actual))
        with pytest.raises(M.MpcError):
            solver.history_update(samples[:5])                                       # not the fleet the rings were made for
    finally:
        solver.close()


def _with_current(p, c):
    import copy
    q = copy.copy(p)
    q = type(p).from_buffer_copy(bytes(p))
    q.current_x, q.current_y = c
    return q
