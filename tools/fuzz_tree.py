"""A long random run of the tree search against the oracle's DFS (python tools/fuzz_tree.py [problems] [seed]): random
start states, last controls, currents, weights (some zero), granularities and Dubins references; depths 2-4 (the oracle's
depth 4 costs 33 ms); both expansions; further arguments: "hulls" = a random boat per problem, "deep" = through
mpc_hip_solve_tree_deep (with MPC_HIP_TREE_DEEP_FROM=2: levels 3 and 4 in pieces).  Prints every disagreement instead of stopping at the first.  A chosen leaf that differs from the
oracle's counts as a disagreement only if the level does not carry libm_tie (mpc_hip.h: the strict runner-up within 2 ulps:
the choice then depends on whose sin/cos/atan2/exp evaluated it)."""
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import mpc_amd as M
from mpc_amd import workloads as W
from helpers import oracle_args, rel_err, traj_array
import oracle as O


def draw(rng, max_depth=4):
    """the next random problem of the stream -> (problem, depth, weights)"""
    depth = int(rng.randint(2, max_depth + 1))
    speed = float(rng.uniform(0.5, 2.75))
    words = [(int(rng.randint(0, 4)), (float(rng.uniform(0.1, 1.5)), float(rng.uniform(0.5, 4.0)), float(rng.uniform(0.1, 1.5)))),
             (W.LSL, (0.0, 40.0, 0.0))]
    plan = W.chain(words, float(rng.uniform(6.0, 20.0)), speed, (float(rng.uniform(-100, 100)), float(rng.uniform(-100, 100)),
                                                                  float(rng.uniform(0, 2 * math.pi))), W.T0)
    cps = [plan.sample(W.T0 + (i + 1) * 1.0) for i in range(depth)]
    x, y, heading, _ = plan.sample(W.T0)
    start = (x + float(rng.uniform(-3, 3)), y + float(rng.uniform(-3, 3)), (heading + float(rng.uniform(-0.6, 0.6))) % (2 * math.pi),
             float(rng.choice([0.0, rng.uniform(0.0, 2.75), rng.uniform(2.75, 3.5)])))
    w = rng.uniform(0.0, 30.0, size=3) * (rng.uniform(size=3) > 0.15)
    if not w.any():
        w[0] = 1.0
    p = M.make_problem(start, cps, M.substep_schedule(1.0), last=(float(rng.randint(-16, 17)) * 0.0625, float(rng.randint(0, 9)) * 0.125),
                       current=(float(rng.uniform(-0.8, 0.8)), float(rng.uniform(-0.8, 0.8))), weights=tuple(float(v) for v in w))
    p.rudder_granularity = float(rng.choice([0.0625, 0.125, 0.25, 0.03125]))
    p.throttle_granularity = float(rng.choice([0.125, 0.25, 0.0625]))
    return p, depth, w


def draw_hull(rng):
    """another boat (VehicleState.h:86-96): from a heavy barge to a light, overpowered skiff whose speed changes by m/s within
    one level -- what the final level's reach test (mpc_tree.hip: node_margin) has to stay on the safe side of.  Drawn again
    while the reference's 0.1-s Euler step would be unstable for it at the speeds draw() starts from (drag alone reversing
    the speed within a substep: every score NaN within a level or two, and the reference's own result undefined --
    controller.cpp:160 never takes a NaN score, :188 copies a trajectory nobody wrote)."""
    while True:
        h = dict(mass=float(rng.uniform(150.0, 6000.0)), max_power=float(rng.uniform(500.0, 20000.0)),
                 max_speed=float(rng.uniform(1.2, 6.0)), max_rpm=float(rng.uniform(1500.0, 4000.0)),
                 idle_rpm=float(rng.choice([0.0, rng.uniform(0.0, 600.0)])),
                 max_rpm_change_rate=float(rng.uniform(300.0, 4000.0)), max_rudder_angle=float(rng.uniform(0.2, 0.6)),
                 rudder_coefficient=float(rng.uniform(0.05, 0.3)))
        drag_coefficient = h["max_power"] / h["max_speed"] ** 4                      # VehicleState.h:101
        v = max(3.5, h["max_speed"])
        if drag_coefficient * v * v * 0.1 / h["mass"] <= 0.5:
            return h


def run(n_problems=200, seed=7, solver=None, verbose=True, max_depth=4, hulls=False, deep=False):
    """-> dict(problems, bad = [disagreements that do NOT carry libm_tie], libm_ties = [flagged leaf disagreements], worst)
    hulls: every problem with a boat of its own (draw_hull), on the library's side and the oracle's"""
    rng = np.random.RandomState(seed)
    solver = solver or M.Solver(0)
    bad, flagged, worst = [], [], 0.0
    t0 = time.time()
    try:
        return _run(n_problems, rng, solver, verbose, max_depth, hulls, bad, flagged, worst, t0, deep)
    finally:
        if hulls:
            O.set_model()
            solver.set_model(M.default_model())


def _run(n_problems, rng, solver, verbose, max_depth, hulls, bad, flagged, worst, t0, deep=False):
    # deep: through mpc_hip_solve_tree_deep -- with MPC_HIP_TREE_DEEP_FROM=2 in the environment (and the switches that send
    # shallow levels the synchronised way and through list and ghosts) levels 3 and 4 are then searched in pieces
    solve = solver.solve_tree_deep if deep else solver.solve_tree
    for k in range(n_problems):
        p, depth, w = draw(rng, max_depth)
        if hulls:
            hull = draw_hull(rng)
            O.set_model(**hull)
            solver.set_model(M.default_model(**hull))
        cfg, veh, ocps, sub = oracle_args(p)
        it, traj, olev, nodes = O.mpc_tree(cfg, O.State(p.start_x, p.start_y, p.start_heading, p.start_speed, 0.0), ocps, depth, -1)
        results = []
        for prec in (M.PREC_F64, M.PREC_FAST):
            try:
                res, levels = solve(p, depth, precision=prec)
            except M.MpcError as e:
                # a drawn hull that turns faster than the fast model's polynomials cover: the library says so, strict only
                if hulls and prec == M.PREC_FAST and "does not cover this hull" in str(e):
                    results.append(results[0])
                    continue
                raise
            why = []
            if res.status != 0 or res.levels != it:
                why.append("status/levels %d %d vs %d" % (res.status, res.levels, it))
            deepest_is_a_flagged_tie = False
            for l in range(min(it, res.levels)):
                if levels[l].n_leaves != olev[l].n_leaves: why.append("leaves at level %d: %d vs %d" % (l, levels[l].n_leaves, olev[l].n_leaves))
                if levels[l].best_leaf != olev[l].best_leaf:
                    what = "leaf at level %d: %d vs %d (scores %.17g %.17g)" % (l, levels[l].best_leaf, olev[l].best_leaf, levels[l].best_score, olev[l].best_score)
                    # another leaf than the oracle's is admissible only where the result says the choice is libm-dependent
                    if levels[l].libm_tie:
                        flagged.append((k, prec, depth, what))
                        deepest_is_a_flagged_tie |= l == res.levels - 1
                    else:
                        why.append(what)
                worst = max(worst, float(rel_err(levels[l].best_score, olev[l].best_score)))
            got = traj_array(res)
            for l in range(min(it, res.levels) + 1):
                if deepest_is_a_flagged_tie:
                    break                     # the trajectory is the other leaf's
                want = [traj[l].v.s.x, traj[l].v.s.y, traj[l].v.s.heading, traj[l].v.s.speed, traj[l].v.rpm]
                e = float(rel_err(got[l, :5], want).max())
                if e > 1e-12: why.append("trajectory level %d off by %.3g" % (l, e))
                if got[l, 7] != traj[l].last_rudder or got[l, 8] != traj[l].last_throttle: why.append("controls at level %d" % l)
            if why:
                bad.append((k, prec, depth, tuple(float(v) for v in w), why[:3]))
            results.append((res.best_ordinal, res.best_last, res.best_cumulative, res.libm_tie, traj_array(res).tobytes(),
                            tuple(levels[l].libm_tie for l in range(res.levels))))
        if results[0] != results[1]:
            bad.append((k, "fast expansion differs from the strict one", depth))
        if verbose and k % 50 == 49:
            print("%d problems, %d unflagged disagreements, %d flagged libm ties, worst level-score error %.3g, %.0f s"
                  % (k + 1, len(bad), len(flagged), worst, time.time() - t0), flush=True)
    return {"problems": n_problems, "bad": bad, "libm_ties": flagged, "worst": worst}


if __name__ == "__main__":
    out = run(int(sys.argv[1]) if len(sys.argv) > 1 else 200, int(sys.argv[2]) if len(sys.argv) > 2 else 7,
              hulls="hulls" in sys.argv[3:], deep="deep" in sys.argv[3:])
    print("tree problems", out["problems"], "; disagreements WITHOUT the libm_tie flag", len(out["bad"]),
          "; leaf choices that differ from the oracle's and carry libm_tie", len(out["libm_ties"]),
          "; worst relative error of a level's best score", out["worst"])
    for b in out["bad"]:
        print("   ", b)
    for b in out["libm_ties"]:
        print("    flagged:", b)
