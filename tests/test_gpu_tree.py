"""GPU parity of the reference-native limited-branching tree search (controller.cpp:80-189) against the oracle's
DFS restatement: per deepening iteration the leaf count, the DFS ordinal of the chosen leaf and its scores, and
the returned trajectory / first control.  Scenarios follow the reference's test_controller.cpp: the LRL
semicircle of updateReferenceTrajectoryTest (:449-467) and the go-straight / turn scenarios of the
commented-out tests (:44-267)."""
import math
import os
import sys

import numpy as np
import pytest

import mpc_amd as M
from mpc_amd import workloads as W
from helpers import oracle_args, rel_err, traj_array
import oracle as O

pytestmark = pytest.mark.gpu
TOL = 1e-12


def _oracle_tree(p, max_levels, budget=-1):
    cfg, veh, cps, sub = oracle_args(p)
    start = O.State(p.start_x, p.start_y, p.start_heading, p.start_speed, 0.0)
    return O.mpc_tree(cfg, start, cps, max_levels, budget)


def _check_one(solver, p, max_levels, prec):
    res, levels = solver.solve_tree(p, max_levels, precision=prec)
    it, traj, olev, nodes = _oracle_tree(p, max_levels)
    assert res.status == 0 and res.levels == it == max_levels
    for l in range(it):
        assert levels[l].n_leaves == olev[l].n_leaves, l
        assert levels[l].best_leaf == olev[l].best_leaf, (prec, l)
        assert rel_err(levels[l].best_score, olev[l].best_score) <= TOL
        assert rel_err(levels[l].best_cumulative, olev[l].best_cumulative) <= TOL
    got = traj_array(res)
    for l in range(it + 1):
        want = [traj[l].v.s.x, traj[l].v.s.y, traj[l].v.s.heading, traj[l].v.s.speed, traj[l].v.rpm]
        assert rel_err(got[l, :5], want).max() <= TOL, l
        assert got[l, 7] == traj[l].last_rudder and got[l, 8] == traj[l].last_throttle, l
        assert rel_err(got[l, 9], traj[l].score) <= TOL
        if l:
            assert rel_err(got[l, 5:7], [traj[l].v.cmg, traj[l].v.sog]).max() <= 1e-10
    assert res.first_rudder == traj[1].last_rudder and res.first_throttle == traj[1].last_throttle
    assert res.evaluated == sum(olev[l].n_leaves for l in range(it))
    return res, levels


def _check(solver, p, max_levels):
    """strict expansion, then the fast expansion (streamlined model + strict decision among near-tied leaves + the
    winner's strict path): the same leaf at every level and bit-identical scores and trajectory"""
    res, levels = _check_one(solver, p, max_levels, M.PREC_F64)
    fres, flevels = _check_one(solver, p, max_levels, M.PREC_FAST)
    assert traj_array(fres).tobytes() == traj_array(res).tobytes()
    assert (fres.best_ordinal, fres.best_last, fres.best_cumulative) == (res.best_ordinal, res.best_last, res.best_cumulative)
    for a, b in zip(flevels[:max_levels], levels[:max_levels]):
        assert (a.n_leaves, a.best_leaf, a.best_score, a.best_cumulative) == (b.n_leaves, b.best_leaf, b.best_score, b.best_cumulative)
        assert a.libm_tie == b.libm_tie
    assert fres.libm_tie == res.libm_tie
    return res


def _semicircle(length_s, current=(0.2, -0.1), weights=(1.0, 20.0, 5.0)):
    """test_controller.cpp:449-467: LRL, rho 8, v 2.5, start t = 4, truncated to length_s seconds of plan."""
    w = W.DubinsWrapper((0.0, 0.0, math.pi / 2), (0.0, math.pi, 0.0), 8.0, W.LRL, 2.5, 4.0)
    w.end_time = 4.0 + length_s
    plan = W.DubinsPlan([w])
    depth = plan.max_iterations(4.0)
    return M.make_problem((0.0, 0.0, 0.0, 2.5), plan.checkpoints(4.0, depth), M.substep_schedule(1.0),
                          last=(0.0, 0.0), current=current, weights=weights), depth


@pytest.mark.parametrize("length_s", [3.5, 4.5, 5.5])
def test_semicircle_matches_oracle_and_survey_probe(solver, length_s):
    p, depth = _semicircle(length_s)
    assert depth == int(length_s - 1)
    res = _check(solver, p, depth)
    # SURVEY.md section 8c probe of the compiled reference: the chosen first control
    want = {3.5: (0.0625, 1.0), 4.5: (-0.0625, 1.0), 5.5: (-0.0625, 1.0)}[length_s]
    assert (res.first_rudder, res.first_throttle) == want


def test_semicircle_reference_test_weights(solver):
    """updateConfig(0.0625, 0.125, 1, 1, 0, ...) of test_controller.cpp:455"""
    p, depth = _semicircle(5.5, current=(0.0, 0.0), weights=(1.0, 1.0, 0.0))
    _check(solver, p, depth)


@pytest.mark.parametrize("depth", [3, 4, 5])
def test_config1_straight_line(solver, depth):
    p, plan = W.config1_problem(depth=depth)
    res = _check(solver, p, depth)
    if depth <= 4:
        assert res.first_rudder == 0.0          # on the line, heading along it: no rudder
    # at depth 5 the reference itself picks rudder 1.0: only the LAST level's score is compared
    # (controller.cpp:158-166), and _check has shown the GPU makes the same choice


@pytest.mark.parametrize("name,start,last", [
    ("north_from_rest", (0.0, 0.0, 0.0, 0.0), (0.0, 0.0)),
    ("east_start_north", (0.0, 0.0, math.pi / 2, 2.0), (0.5, 0.75)),
    ("turn_around", (0.0, 0.0, math.pi, 2.5), (-0.3125, 0.625)),
    ("offset_last_near_bounds", (3.0, -2.0, 0.2, 1.5), (0.96875, 0.0625)),
])
def test_scenarios(solver, name, start, last):
    # reference heading north along x = 0 from (0, 0), 2.5 m/s: LSL with only the straight segment
    plan = W.DubinsPlan([W.DubinsWrapper((0.0, 0.0, math.pi / 2), (0.0, 5.0, 0.0), 8.0, W.LSL, 2.5, 10.0)])
    depth = 4
    p = M.make_problem(start, plan.checkpoints(10.0, depth), M.substep_schedule(1.0), last=last, current=(0.1, 0.05))
    _check(solver, p, depth)


def test_self_generated_reference_is_recovered(solver):
    """realStateTest (test_controller.cpp:174-209): the reference is the model's own response to a constant
    control; MPC must pick a first control that reproduces it."""
    cfg = O.default_config()
    v = O.vehicle(0, 0, 0, 2.0, 0.0)
    cps = []
    for i in range(4):
        v = O.simulate(v, 0.0625, 0.875, 1.0)
        cps.append((v.s.x, v.s.y, v.s.heading, v.s.speed))
    p = M.make_problem((0, 0, 0, 2.0), cps, M.substep_schedule(1.0), last=(0.0625, 0.875))
    res = _check(solver, p, 4)
    assert (res.first_rudder, res.first_throttle) == (0.0625, 0.875)


def test_deadline_and_node_budget(solver):
    p, depth = _semicircle(6.5)
    # frozen clock already past the deadline: no iteration completes -> controller.cpp:196
    with pytest.raises(M.MpcError) as e:
        solver.solve_tree(p, depth, now=lambda: 10.0, end_time=5.0)
    assert e.value.status == 5 and "failed to complete a single iteration" in str(e.value)
    # The clock passes the deadline after a fixed number of polls: the deepest COMPLETED level is returned.  Polls: one
    # before and one after the shallow batch (levels 1-4, expanded without host round trips), then before and after
    # every further level; a level that completes after the deadline is discarded (controller.cpp:101-103).
    for polls_before_deadline, want_levels in ((2, 4), (3, 4), (4, 5)):
        calls = {"n": 0}

        def clock():
            calls["n"] += 1
            return 0.0 if calls["n"] <= polls_before_deadline else 100.0
        res, levels = solver.solve_tree(p, depth, now=clock, end_time=50.0)
        assert res.levels == want_levels, (polls_before_deadline, res.levels)
        it, traj, olev, nodes = _oracle_tree(p, want_levels)
        assert levels[want_levels - 1].best_leaf == olev[want_levels - 1].best_leaf
        assert res.first_rudder == traj[1].last_rudder and res.first_throttle == traj[1].last_throttle
    # a real clock whose deadline falls INSIDE the shallow batch: the levels whose reduction the GPU's own clock places
    # before the deadline count; however many that is on this box, the answer is that depth's search
    import time
    for budget in (20e-6, 60e-6, 150e-6, 400e-6):
        t0 = time.perf_counter()
        try:
            res, levels = solver.solve_tree(p, 4, now=time.perf_counter, end_time=t0 + budget)
        except M.MpcError as e:
            assert e.status == 5
            continue
        assert 1 <= res.levels <= 4
        it, traj, olev, nodes = _oracle_tree(p, res.levels)
        assert levels[res.levels - 1].best_leaf == olev[res.levels - 1].best_leaf
        assert res.first_rudder == traj[1].last_rudder and res.first_throttle == traj[1].last_throttle
    # a level wider than max_nodes ends the search the same way
    res, levels = solver.solve_tree(p, depth, max_nodes=1000)
    assert res.levels == 2                                   # level 3 has 3745 leaves


@pytest.mark.parametrize("prec", [M.PREC_F64, M.PREC_FAST])
@pytest.mark.parametrize("n_shards,split", [(2, 1), (3, 2), (8, 2)])
def test_sharded_search_equals_the_whole_search(solver, prec, n_shards, split):
    """mpc_hip_solve_tree_shard (tree mode): the shards of one GPU-count run one after the other on this
    GPU; combined, every level's leaf count, winner (global DFS ordinal) and score are the unsharded search's, and the
    owning shard returns the same trajectory."""
    import torch
    from mpc_amd.distributed import tree_levels_tensor, combine_tree_levels
    p, depth = _semicircle(6.5)
    depth = min(depth, 5)
    whole, wl = solver.solve_tree(p, depth, precision=prec)
    shards = [solver.solve_tree(p, depth, precision=prec, split_level=split, shard=s, n_shards=n_shards) for s in range(n_shards)]
    gathered = torch.stack([tree_levels_tensor(lv, depth) for _, lv in shards])
    combined = combine_tree_levels(gathered, split)
    for l in range(depth):
        total, ordinal, bits, owner = combined[l]
        assert total == wl[l].n_leaves and ordinal == wl[l].best_leaf, (l, combined[l])
        assert bits == int(np.float64(wl[l].best_score).view(np.int64))
    owner = combined[-1][3]
    assert traj_array(shards[owner][0]).tobytes() == traj_array(whole).tobytes()
    assert sum(r.evaluated for r, _ in shards) >= whole.evaluated           # the levels above the split are redundant
    with pytest.raises(M.MpcError):
        solver.solve_tree(p, depth, split_level=depth, shard=0, n_shards=2)  # the split level must be a stored level


@pytest.mark.parametrize("prec", [M.PREC_F64, M.PREC_FAST])
@pytest.mark.parametrize("how", ["deadline", "max_nodes"])
def test_sharded_search_that_ends_at_its_split_level(solver, prec, how):
    """A shard > 0 whose search stops right after the split level (the clock passes end_time, or the next level's slice
    is wider than max_nodes) has already sliced that level: its result must still be the WHOLE level's winner -- the
    trajectory of the unsharded search truncated there -- not the node `shard offset + leaf` of the slice."""
    p, depth = _semicircle(6.5)
    depth, split = min(depth, 5), 2
    want, wl = solver.solve_tree(p, split, precision=prec)                 # the unsharded search, truncated at the split level
    for shard, n_shards in ((1, 2), (2, 3)):
        if how == "deadline":
            calls = [0]

            def now():                       # polled before and after the shallow batch (here: up to the split level), then per level
                calls[0] += 1
                return 0.0 if calls[0] <= 2 else 2.0
            res, lv = solver.solve_tree(p, depth, precision=prec, now=now, end_time=1.0, split_level=split, shard=shard, n_shards=n_shards)
        else:                                # level 3 has 3745 leaves: no shard's slice of it fits 1000 nodes
            res, lv = solver.solve_tree(p, depth, precision=prec, max_nodes=1000, split_level=split, shard=shard, n_shards=n_shards)
        assert res.status == 0 and res.levels == split
        assert (res.best_ordinal, res.best_last, res.best_cumulative) == (want.best_ordinal, want.best_last, want.best_cumulative)
        assert traj_array(res).tobytes() == traj_array(want).tobytes()
        for l in range(split):
            assert (lv[l].n_leaves, lv[l].best_leaf, lv[l].best_score) == (wl[l].n_leaves, wl[l].best_leaf, wl[l].best_score)


@pytest.mark.parametrize("case", range(8))
def test_random_tree_problems(solver, case):
    """randomly drawn references, start states, currents, weights, last controls and granularities (fixed seed): the
    same leaf counts, chosen leaves, scores and trajectories as the oracle's DFS at every deepening iteration, from
    both expansions"""
    rng = np.random.RandomState(500 + case)
    depth = int(rng.randint(2, 5))
    words = [(int(rng.randint(0, 6)), (float(rng.uniform(0.1, 1.2)), float(rng.uniform(3.3, 4.5)), float(rng.uniform(0.1, 1.2)))),
             (W.LSL, (0.0, 10.0, 0.0))]
    plan = W.chain(words, float(rng.uniform(6.0, 16.0)), float(rng.uniform(1.0, 2.75)),
                   (float(rng.uniform(-50, 50)), float(rng.uniform(-50, 50)), float(rng.uniform(0, 2 * math.pi))), 10.0)
    x, y, heading, _ = plan.sample(10.0)
    start = (x + float(rng.uniform(-2, 2)), y + float(rng.uniform(-2, 2)), (heading + float(rng.uniform(-0.5, 0.5))) % (2 * math.pi),
             float(rng.uniform(0.0, 3.0)))
    gr, gt = float(rng.choice([0.0625, 0.125, 0.03125])), float(rng.choice([0.125, 0.25, 0.0625]))
    last = (float(rng.randint(-8, 9)) * gr, float(rng.randint(0, int(1 / gt) + 1)) * gt)
    w = rng.uniform(0.5, 25.0, size=3)
    p = M.make_problem(start, plan.checkpoints(10.0, depth), M.substep_schedule(1.0), last=last,
                       current=(float(rng.uniform(-0.6, 0.6)), float(rng.uniform(-0.6, 0.6))), weights=tuple(float(v) for v in w),
                       rudder_granularity=gr, throttle_granularity=gt)
    _check(solver, p, depth)


def test_the_first_level_beyond_the_storage_cap_is_the_final_one(solver):
    """no node budget and more levels asked for than fit: levels are stored up to the library's cap (2^28 nodes) and the
    first level beyond it -- depth 7, 3.0e8 leaves -- is expanded without stores; the answer is the depth-7 search's"""
    q, _plan = W.config1_problem(depth=8)
    assert q.depth == 8
    for prec in (M.PREC_FAST, M.PREC_F64):
        deep, dl = solver.solve_tree(q, 8, precision=prec)
        ref, rl = solver.solve_tree(q, 7, precision=prec)
        assert deep.levels == ref.levels == 7 and dl[6].n_leaves == 303370833
        assert (deep.best_ordinal, deep.best_last, deep.best_cumulative) == (ref.best_ordinal, ref.best_last, ref.best_cumulative)
        assert traj_array(deep).tobytes() == traj_array(ref).tobytes()


def _control_sets(p, last_r, last_t):
    """controller.cpp:121-146 in Python floats (the same IEEE operations): the control set of a node from its incoming control"""
    tol = p.tolerance
    rud = [p.min_rudder, p.mid_rudder, p.max_rudder]
    lo, hi = last_r - p.rudder_granularity, last_r + p.rudder_granularity
    if p.min_rudder + tol < lo and abs(p.mid_rudder - lo) > tol:
        rud.append(lo)
    if abs(p.mid_rudder - hi) > tol and p.max_rudder > hi + tol:
        rud.append(hi)
    if abs(p.min_rudder - last_r) > tol and abs(p.mid_rudder - last_r) > tol and abs(p.max_rudder - last_r) > tol:
        rud.append(last_r)
    thr = [p.min_throttle, p.max_throttle]
    lo, hi = last_t - p.throttle_granularity, last_t + p.throttle_granularity
    if p.min_throttle + tol < lo:
        thr.append(lo)
    if p.max_throttle > hi + tol:
        thr.append(hi)
    if abs(p.min_throttle - last_t) > tol and abs(p.max_throttle - last_t) > tol:
        thr.append(last_t)
    return rud, thr


def _leaf_counts(p, depth):
    """leaves per level without simulating anything: the tree's shape depends on the controls alone"""
    states = {(p.last_rudder, p.last_throttle): 1}
    counts = []
    for _ in range(depth):
        nxt = {}
        for (r, t), mult in states.items():
            rud, thr = _control_sets(p, r, t)
            for tt in thr:
                for rr in rud:
                    nxt[(rr, tt)] = nxt.get((rr, tt), 0) + mult
        states = nxt
        counts.append(sum(states.values()))
    return counts


def test_depth_8_as_the_8_shards_of_a_node(solver):
    """The reference's deepening loop has no depth cap but its deadline (controller.cpp:80,101).  Depth 8 -- 5.2e9
    leaves, 5.7e10 simulate substeps -- is what a node's 8 GPUs buy the production path: each shard's leaves fit 32-bit
    local ordinals, the global DFS ordinal is 64-bit.  The 8 shards run one after the other on this GPU.  Checked: per
    level the shards' leaf counts sum to the count recursion of the control sets; levels 1-4 equal the oracle's DFS;
    the winner's path re-scored by the oracle's rollout matches the returned trajectory to 1e-12; no sampled leaf
    (random valid control sequences, scored by the oracle) beats the winner."""
    import torch
    from mpc_amd.distributed import tree_levels_tensor, combine_tree_levels
    depth, n_shards, split = 8, 8, 3
    p, _plan = W.config1_problem(depth=depth)
    assert p.depth == depth
    shards = [solver.solve_tree(p, depth, precision=M.PREC_FAST, split_level=split, shard=s, n_shards=n_shards)
              for s in range(n_shards)]
    assert all(r.status == 0 and r.levels == depth for r, _ in shards)
    combined = combine_tree_levels(torch.stack([tree_levels_tensor(lv, depth) for _, lv in shards]), split)
    want_counts = _leaf_counts(p, depth)
    assert [c[0] for c in combined] == want_counts and want_counts[-1] == 5172889071 > 2 ** 32
    # the shallow levels against the oracle's own DFS
    it, traj, olev, _ = _oracle_tree(p, 4)
    for l in range(4):
        assert (combined[l][0], combined[l][1]) == (olev[l].n_leaves, olev[l].best_leaf), l
        assert combined[l][2] == int(np.float64(olev[l].best_score).view(np.int64)) or \
            rel_err(np.int64(combined[l][2]).view(np.float64), olev[l].best_score) <= TOL
    total, ordinal, bits, owner = combined[-1]
    assert ordinal >= 0 and total == want_counts[-1]
    res = shards[owner][0]
    got = traj_array(res)
    assert rel_err(res.best_last, np.int64(bits).view(np.float64)) == 0.0
    # the winner's path through the oracle
    cfg, veh, cps, sub = oracle_args(p)
    o = O.rollout(cfg, veh, cps, depth, sub, got[1:, 7], got[1:, 8])
    for l in range(depth + 1):
        want = [o[l].v.s.x, o[l].v.s.y, o[l].v.s.heading, o[l].v.s.speed, o[l].v.rpm]
        assert rel_err(got[l, :5], want).max() <= TOL, l
        assert rel_err(got[l, 9], o[l].score) <= TOL
    best_last = o[depth].score - o[depth - 1].score
    assert rel_err(res.best_last, best_last) <= 1e-10                 # a difference of two cumulative scores
    # the path is a valid one: every control comes from its parent's control set
    lr, lt = p.last_rudder, p.last_throttle
    for l in range(1, depth + 1):
        rud, thr = _control_sets(p, lr, lt)
        assert got[l, 7] in rud and got[l, 8] in thr, l
        lr, lt = got[l, 7], got[l, 8]
    # sampled leaves: none has a smaller last-step score
    rng = np.random.RandomState(8)
    for _ in range(400):
        lr, lt, rs, ts = p.last_rudder, p.last_throttle, [], []
        for l in range(depth):
            rud, thr = _control_sets(p, lr, lt)
            lr, lt = rud[rng.randint(len(rud))], thr[rng.randint(len(thr))]
            rs.append(lr); ts.append(lt)
        s = O.rollout(cfg, veh, cps, depth, sub, rs, ts)
        assert s[depth].score - s[depth - 1].score >= res.best_last * (1 - 1e-10)


@pytest.mark.parametrize("prec", [M.PREC_FAST, M.PREC_F64])
def test_deepening_beyond_one_expansion_equals_the_shards(solver, prec):
    """mpc_hip_solve_tree_deep: depth 8 (5.2e9 leaves: beyond 32-bit ordinals and the storage cap) on one handle, in pieces
    -- the same winner, scores, trajectory and leaf count as the 8 shards of test_depth_8_as_the_8_shards_of_a_node
    combined, the shallower levels those of the plain search; with a clock that runs out among the pieces the
    unfinished level does not count (controller.cpp:101-103) and the result is the plain search's depth 7."""
    import torch
    from mpc_amd.distributed import tree_levels_tensor, combine_tree_levels
    depth = 8
    p, _plan = W.config1_problem(depth=depth)
    plain, pl = solver.solve_tree(p, depth, precision=prec)
    assert plain.status == 0 and plain.levels == 7                     # where one expansion ends
    deep, dl = solver.solve_tree_deep(p, depth, precision=prec)
    assert deep.status == 0 and deep.levels == 8
    for l in range(7):
        assert (dl[l].n_leaves, dl[l].best_leaf, dl[l].best_score, dl[l].libm_tie) == (pl[l].n_leaves, pl[l].best_leaf, pl[l].best_score, pl[l].libm_tie)
    shards = [solver.solve_tree(p, depth, precision=prec, split_level=3, shard=s, n_shards=8) for s in range(8)]
    total, ordinal, bits, owner = combine_tree_levels(torch.stack([tree_levels_tensor(lv, depth) for _, lv in shards]), 3)[-1]
    assert (dl[7].n_leaves, dl[7].best_leaf) == (total, ordinal) and total == 5172889071
    assert int(np.float64(dl[7].best_score).view(np.int64)) == bits
    assert deep.best_ordinal == ordinal and deep.best_last == dl[7].best_score
    assert traj_array(deep).tobytes() == traj_array(shards[owner][0]).tobytes()
    assert deep.evaluated == sum(dl[l].n_leaves for l in range(8))
    # the clock: 1 s per poll.  Counted first (a deadline far away): the plain search polls a handful of times, the pieces'
    # searches many more; then a deadline half way between the two runs out among the pieces
    t = [0.0]
    def clock():
        t[0] += 1.0
        return t[0]
    solver.solve_tree(p, depth, precision=prec, now=clock, end_time=1e9)
    polls_plain, t[0] = t[0], 0.0
    full, _ = solver.solve_tree_deep(p, depth, precision=prec, now=clock, end_time=1e9)
    polls_deep, t[0] = t[0], 0.0
    assert full.levels == 8 and polls_deep >= polls_plain + 4
    cut, cl = solver.solve_tree_deep(p, depth, precision=prec, now=clock, end_time=0.5 * (polls_plain + polls_deep))
    assert cut.status == 0 and cut.levels == 7, (cut.levels, t[0], polls_plain, polls_deep)
    assert traj_array(cut).tobytes() == traj_array(plain).tobytes() and cut.best_ordinal == plain.best_ordinal


def test_ghosts_and_short_list_change_no_result():
    """the level above a final level keeps only a place (no state) for nodes that cannot matter (mpc_tree.hip: ghosts) and the
    final level is expanded from a short list of its tiles: two child processes search the depth-8 tree (5.2e9 leaves, in
    pieces), one with the library's diagnostic switch that turns the ghosts off -- same levels, leaves, first control,
    ordinal and tie flag"""
    import subprocess
    probe = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "gpu_probe_deep.py")
    lines = []
    for ghosts in ("1", "0"):
        r = subprocess.run([sys.executable, "-u", probe, "fast", "8", "1"], env=dict(os.environ, MPC_HIP_TREE_GHOSTS=ghosts),
                           stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=200)
        assert r.returncode == 0, r.stdout[-2000:]
        line = [l for l in r.stdout.splitlines() if " deep depth=8 " in l][-1]
        lines.append(" ".join(w for w in line.split() if not w.startswith("wall_ms=")))
    assert lines[0] == lines[1] and "levels=8 leaves=5172889071" in lines[0], lines


def _leaf_controls(p, depth):
    """every leaf of the depth-`depth` tree in DFS order (throttle outer, rudder inner: controller.cpp:174-182) as control
    sequences [depth][n_leaves]"""
    paths = [([], [], p.last_rudder, p.last_throttle)]
    for _ in range(depth):
        nxt = []
        for rs, ts, lr, lt in paths:
            rud, thr = _control_sets(p, lr, lt)
            for tt in thr:
                for rr in rud:
                    nxt.append((rs + [rr], ts + [tt], rr, tt))
        paths = nxt
    return (np.array([q[0] for q in paths], dtype=np.float64).T.copy(), np.array([q[1] for q in paths], dtype=np.float64).T.copy())


@pytest.mark.parametrize("which", ["semicircle", "straight", "straight_with_current", "speed_only", "random"])
def test_libm_tie_flag_is_the_two_ulp_test_on_the_strict_scores(solver, which):
    """mpc_tree_level.libm_tie / mpc_result.libm_tie (mpc_hip.h): set exactly when a level's strict runner-up lies within
    MPC_TIE_REL = 1e-12 of its winner, relative to max(score, 1 % of the weights' sum).  The expectation comes from the
    device's own strict scores of EVERY leaf: the
    tree's leaves, enumerated on the host in DFS order, evaluated as an explicit candidate set in MPC_PREC_F64."""
    import torch
    depth = 3
    if which == "semicircle":
        p, _ = _semicircle(4.5)
    elif which == "straight":
        p, _ = W.config1_problem(depth=depth)
    elif which == "straight_with_current":
        p, _ = W.config1_problem(depth=depth)
        p.current_x, p.current_y = 0.15, -0.1
    elif which == "speed_only":
        # only the speed term is weighted and there is no current: speed over ground does not depend on the heading but
        # for its rounding, so leaves that differ in their rudders alone score within an ulp or two of each other
        p, _ = W.config1_problem(depth=depth)
        p.w_distance, p.w_heading, p.w_speed = 0.0, 0.0, 5.0
    else:
        rng = np.random.RandomState(77)
        plan = W.chain([(W.RSL, (0.4, 3.5, 0.7)), (W.LSL, (0.0, 10.0, 0.0))], 9.0, 2.2, (5.0, -3.0, 1.0), 10.0)
        x, y, heading, _ = plan.sample(10.0)
        p = M.make_problem((x + 1.0, y - 0.5, heading + 0.2, 1.7), plan.checkpoints(10.0, depth), M.substep_schedule(1.0),
                           last=(0.1875, 0.625), current=(0.2, 0.1), weights=tuple(float(v) for v in rng.uniform(0.5, 25.0, size=3)))
    strict, sl = solver.solve_tree(p, depth, precision=M.PREC_F64)
    fast, fl = solver.solve_tree(p, depth, precision=M.PREC_FAST)
    seen = []
    for l in range(depth):
        rud, thr = _leaf_controls(p, l + 1)
        n = rud.shape[1]
        assert n == sl[l].n_leaves
        q = M.make_problem((p.start_x, p.start_y, p.start_heading, p.start_speed),
                           [(p.checkpoints[i].x, p.checkpoints[i].y, p.checkpoints[i].heading, p.checkpoints[i].speed) for i in range(l + 1)],
                           [p.substeps[i] for i in range(p.n_substeps)], last=(p.last_rudder, p.last_throttle),
                           current=(p.current_x, p.current_y), weights=(p.w_distance, p.w_heading, p.w_speed))
        last = torch.empty(n, dtype=torch.float64, device="cuda")
        res = solver.eval_candidates(q, torch.tensor(rud, dtype=torch.float32, device="cuda").contiguous(),
                                     torch.tensor(thr, dtype=torch.float32, device="cuda").contiguous(), n, M.PREC_F64,
                                     M.SELECT_LAST_STEP, last, None)
        c = np.sort(last.cpu().numpy())
        assert c[0] == sl[l].best_score and res.best_ordinal == sl[l].best_leaf
        want = _tie_expected(p, c)
        assert sl[l].libm_tie == want and fl[l].libm_tie == want, (which, l, float(c[1] - c[0]))
        assert res.libm_tie == want                                # the candidate-set path reports the same flag
        seen.append(want)
    assert strict.libm_tie == fast.libm_tie == seen[-1]
    print(which, "libm_tie per level:", seen)
    if which == "speed_only":
        assert any(seen)          # the positive case is really exercised (observed: [1, 0, 1])


def _tie_expected(p, sorted_costs):
    """mpc_tie_limit of mpc_internal.h in the same IEEE operations: runner-up <= m + max(m, 1 % of the weights' sum) * 1e-12"""
    m = np.float64(sorted_costs[0])
    floor = np.float64(0.01) * (np.float64(p.w_distance) + np.float64(p.w_heading) + np.float64(p.w_speed))
    return int(sorted_costs[1] <= m + (m if m > floor else floor) * np.float64(1.0e-12))


def test_libm_tie_of_a_candidate_set_with_a_duplicated_winner(solver):
    """a candidate set in which the winner occurs twice: libm_tie in both precisions (equal scores: 0 ulps apart), the
    lower ordinal wins; without the duplicate the flag follows the MPC_TIE_REL test on the strict costs"""
    import torch
    p = W.config2_problem(depth=8)
    n = 5000
    rud, thr = W.seeded_controls(W.SEED, 0, n, 8, p.last_rudder, p.last_throttle)
    d_r = torch.tensor(rud, dtype=torch.float32, device="cuda").contiguous()
    d_t = torch.tensor(thr, dtype=torch.float32, device="cuda").contiguous()
    last = torch.empty(n, dtype=torch.float64, device="cuda")
    base = solver.eval_candidates(p, d_r, d_t, n, M.PREC_F64, M.SELECT_LAST_STEP, last, None)
    assert base.libm_tie == _tie_expected(p, np.sort(last.cpu().numpy()))
    assert solver.eval_candidates(p, d_r, d_t, n, M.PREC_FAST).libm_tie == base.libm_tie
    w = int(base.best_ordinal)
    for other in (w + 1 if w + 1 < n else w - 1, (w + 2500) % n):          # in the winner's own workgroup, and in another
        r2, t2 = d_r.clone(), d_t.clone()
        r2[:, other] = d_r[:, w]; t2[:, other] = d_t[:, w]
        for prec in (M.PREC_F64, M.PREC_FAST):
            res = solver.eval_candidates(p, r2, t2, n, prec)
            assert res.libm_tie == 1 and res.best_ordinal == min(w, other), (other, prec)
