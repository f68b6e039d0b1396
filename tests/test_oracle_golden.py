"""The oracle pinned against outputs of the reference itself (tests/golden/survey_probe.json: the survey's probe of
the unmodified reference sources) and against the properties the reference's own live tests assert
(test/test_controller.cpp:384-433, :449-467).  CPU only."""
import json
import math
import os

import numpy as np

import oracle as O

G = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "survey_probe.json")))


def ulps(a, b):
    ia = np.array([a], dtype=np.float64).view(np.int64)[0]
    ib = np.array([b], dtype=np.float64).view(np.int64)[0]
    return abs(int(ia) - int(ib))


def test_simulate_golden_vector_is_bit_exact():
    g = G["simulate"]
    v = O.vehicle(*g["start"])
    r = O.simulate(v, g["rudder"], g["throttle"], g["t"], *g["current"])
    got = dict(x=r.s.x, y=r.s.y, heading=r.s.heading, speed=r.s.speed, rpm=r.rpm, cmg=r.cmg, sog=r.sog, time=r.s.time)
    for k, val in got.items():
        assert val == g[k], (k, repr(val), repr(g[k]))


def test_substep_schedule_is_floating_point_defined():
    s1 = O.schedule(1.0)
    assert len(s1) == G["schedule"]["1.0"]["count"] == 11
    assert np.all(s1[:10] == 0.1)
    assert abs(s1[10] / G["schedule"]["1.0"]["last"] - 1) < 1e-4 and 0 < s1[10] < 2e-16
    s2 = O.schedule(2.0)
    assert len(s2) == 20 and s2[-1] == G["schedule"]["2.0"]["last"]
    assert len(O.schedule(5.0)) == G["schedule"]["5.0"]["count"]
    assert len(O.schedule(0.05)) == 1 and O.schedule(0.05)[0] == 0.05


def _tree_case(seconds):
    t = G["tree"]
    pl = t["plan"]
    w = O.dubins(pl["qi"], pl["param"], pl["rho"], O.LRL, pl["speed"], pl["start_time"], end_time=pl["start_time"] + seconds)
    plan = O.plan_array([w])
    start = O.State(*t["start"])
    mi = O.lib().orc_max_iterations(plan, 1, start.time)
    cfg = O.default_config(current_x=t["current"][0], current_y=t["current"][1])
    cps = O.checkpoints(plan, start.time, mi)
    return (mi,) + O.mpc_tree(cfg, start, cps, mi)


def test_tree_search_matches_the_reference_probe():
    for case in G["tree"]["cases"]:
        mi, it, traj, levels, nodes = _tree_case(case["plan_seconds"])
        assert mi == it == case["levels"]
        assert traj[1].last_rudder == case["rudder"] and traj[1].last_throttle == case["throttle"]
        assert ulps(traj[1].score, case["score"]) <= G["tree"]["score_ulps"]
        if "x" in case:
            assert traj[1].v.s.x == case["x"] and traj[1].v.s.y == case["y"]


def test_probe_score_is_the_reassociated_sum():
    """The 1-ulp difference above is the -funsafe-math-optimizations reassociation, nothing else."""
    case = G["tree"]["cases"][1]
    mi, it, traj, levels, nodes = _tree_case(case["plan_seconds"])
    t = G["tree"]; pl = t["plan"]
    w = O.dubins(pl["qi"], pl["param"], pl["rho"], O.LRL, pl["speed"], pl["start_time"], end_time=pl["start_time"] + case["plan_seconds"])
    ref = O.checkpoints(O.plan_array([w]), 4.0, 1)[0]
    v = traj[1].v
    hd = abs(O.lib().orc_heading_difference(ref.heading, v.cmg)); sd = abs(ref.speed - v.sog)
    d = math.exp(math.sqrt((ref.x - v.s.x) ** 2 + (ref.y - v.s.y) ** 2))
    assert 1.0 * d + (20.0 * hd + 5.0 * sd) == case["score"]
    assert 1.0 * d + 20.0 * hd + 5.0 * sd == traj[1].score


def test_iterative_deepening_counts():
    mi, it, traj, levels, nodes = _tree_case(5.5)
    assert [levels[i].n_leaves for i in range(4)] == [15, 230, 3745, 62375]
    # every deepening pass re-simulates all shallower levels (controller.cpp:80-100)
    assert nodes == 15 + (15 + 230) + (15 + 230 + 3745) + (15 + 230 + 3745 + 62375)


def test_node_budget_is_the_deadline():
    t = G["tree"]; pl = t["plan"]
    w = O.dubins(pl["qi"], pl["param"], pl["rho"], O.LRL, pl["speed"], pl["start_time"], end_time=9.5)
    plan = O.plan_array([w]); start = O.State(*t["start"])
    cfg = O.default_config(current_x=0.2, current_y=-0.1)
    cps = O.checkpoints(plan, 4.0, 4)
    it, traj, levels, nodes = O.mpc_tree(cfg, start, cps, 4, node_budget=15 + 245 + 100)
    assert it == 2                      # the third pass ran out of budget and is discarded (controller.cpp:101-103)
    it0, *_ = O.mpc_tree(cfg, start, cps, 4, node_budget=10)
    assert it0 == 0                     # -> "MPC failed to complete a single iteration" (controller.cpp:196)


# ---- the reference's own live tests, as properties of the oracle ------------------------------------------------
def test_reference_estimateTest0_determinism():
    s1 = O.vehicle(0, 0, 2, 2.3, 7)
    a = O.simulate(s1, 0, 1, 3); b = O.simulate(s1, 0, 1, 3)
    assert (a.s.x, a.s.y, a.s.heading, a.s.speed, a.s.time) == (b.s.x, b.s.y, b.s.heading, b.s.speed, b.s.time)


def test_reference_estimateTest1_5s_equals_2s_plus_3s_within_4_ulps():
    s1 = O.vehicle(0, 0, 2, 2.3, 7)
    f = O.simulate(s1, 0, 1, 5)
    s3 = O.simulate(O.simulate(s1, 0, 1, 2), 0, 1, 3)
    for a, b in [(f.s.time, s3.s.time), (f.s.x, s3.s.x), (f.s.y, s3.s.y), (f.s.heading, s3.s.heading), (f.s.speed, s3.s.speed)]:
        assert ulps(a, b) <= G["reference_test_properties"]["estimateTest1_max_ulps"]      # EXPECT_DOUBLE_EQ
    assert ulps(f.s.y, s3.s.y) == 4 and ulps(f.s.x, s3.s.x) == 3 and ulps(f.s.time, s3.s.time) == 2   # SURVEY.md section 4 probe


def test_reference_estimateTest2_partitioning_within_1e3():
    start = O.vehicle(0, 0, 0, 0, 4)
    s2 = O.simulate(O.simulate(start, 0.8, 1, 0.75), 0.8, 1, 0.25)
    s3 = O.simulate(start, 0.8, 1, 1)
    for a, b in [(s2.s.time, s3.s.time), (s2.s.x, s3.s.x), (s2.s.y, s3.s.y), (s2.s.heading, s3.s.heading), (s2.s.speed, s3.s.speed)]:
        assert abs(a - b) <= 1e-3
    assert max(abs(s2.s.x - s3.s.x), abs(s2.s.y - s3.s.y)) > 5e-5       # ... and they do differ (8e-5..6e-4)


def test_reference_semicircle_end_time():
    w = O.dubins((0, 0, math.pi / 2), (0, math.pi, 0), 8, O.LRL, 2.5, 4)
    assert w.end_time == 8 * math.pi / 2.5 + 4            # ASSERT_DOUBLE_EQ, test_controller.cpp:467
    end = O.plan_sample(O.plan_array([w]), w.end_time)     # "do a semi circle": ends at (16, 0) heading south
    assert abs(end.x - 16) < 1e-12 and abs(end.y) < 1e-12 and abs(end.heading - math.pi) < 1e-12


def test_reference_currentEstimatorTest1_returns_zero_as_the_reference_does():
    """The reference's test expects (-0.4, 0) but the code skips its first 20 updates and needs >= 10 samples
    (CurrentEstimator.cpp:7-10,18); the compiled reference returns (0, 0)."""
    start = O.vehicle(0, 0, 0, 0, 1)
    actual = O.simulate(start, -0.8, 1, 1)
    actual.s.x -= 0.4
    assert O.current_estimate([], True, actual.s) == tuple(G["current_estimator_test1"]["reference_returns"])
    assert O.current_estimate([O.HistoryEntry(start.s, -0.8, 1)], True, actual.s) == (0.0, 0.0)


def test_current_estimator_recovers_a_constant_current():
    """>= 10 samples: replay without current, difference / elapsed time (CurrentEstimator.cpp:19-40)."""
    cx, cy = 0.3, -0.15
    v = O.vehicle(5, 5, 1.0, 2.0, 100.0)
    hist = []
    for i in range(30):
        hist.append(O.HistoryEntry(v.s, 0.25, 0.75))
        v = O.simulate(v, 0.25, 0.75, 0.1, cx, cy)
    ex, ey = O.current_estimate(hist, True, v.s)
    assert abs(ex - cx) < 1e-6 and abs(ey - cy) < 1e-6
    assert O.current_estimate(hist, False, v.s) == (0.0, 0.0)           # disabled
    assert O.current_estimate(hist[:9], True, v.s) == (0.0, 0.0)        # < 10 samples
    assert O.current_estimate(hist, True, hist[0].s) == (0.0, 0.0)      # query not after the first sample


def test_heading_convention_and_cmg():
    """headingTowardsTest (test_controller.cpp:13-23): 0 = north, pi/2 = east, clockwise."""
    v = O.simulate(O.vehicle(0, 0, 0.0, 2.0, 0), 0, 0.7, 1.0)
    assert abs(v.s.x) < 1e-12 and v.s.y > 1.5                      # heading 0 moves north (+y)
    v = O.simulate(O.vehicle(0, 0, math.pi / 2, 2.0, 0), 0, 0.7, 1.0)
    assert v.s.x > 1.5 and abs(v.s.y) < 1e-9                       # heading pi/2 moves east (+x)
    assert abs(v.cmg - math.pi / 2) < 1e-12 and abs(v.sog - v.s.speed) < 1e-9   # no current: cmg = heading, sog = speed
    assert abs(O.lib().orc_heading_difference(0.1, 2 * math.pi - 0.1) + 0.2) < 1e-12
