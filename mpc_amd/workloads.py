"""Synthetic workloads of BASELINE.json configs 1-5, product side.

Everything here is INPUT construction: Dubins reference trajectories sampled at the 1-s checkpoints of
controller.cpp:96-98, start states, and the seeded candidate controls of include/mpc_candidates.h as numpy
arrays.  The reference gets these from path_planner_common (DubinsPlan::sample), which is not part of the
reference repository; the closed forms below are A. Walker's Dubins-Curves segment formulas.
tests/ compares this module with the oracle's independent C restatement.
"""
import math

import numpy as np

from . import binding as B

LSL, LSR, RSL, RSR, RLR, LRL = range(6)
_L, _S, _R = 0, 1, 2
_DIRDATA = ((_L, _S, _L), (_L, _S, _R), (_R, _S, _L), (_R, _S, _R), (_R, _L, _R), (_L, _R, _L))
TWO_PI = 2.0 * math.pi


def _mod2pi(a):
    return a - TWO_PI * math.floor(a / TWO_PI)


def _segment(t, qi, kind):
    st, ct = math.sin(qi[2]), math.cos(qi[2])
    if kind == _L:
        q = (math.sin(qi[2] + t) - st, -math.cos(qi[2] + t) + ct, t)
    elif kind == _R:
        q = (-math.sin(qi[2] - t) + st, math.cos(qi[2] - t) - ct, -t)
    else:
        q = (ct * t, st * t, 0.0)
    return (q[0] + qi[0], q[1] + qi[1], q[2] + qi[2])


class DubinsWrapper:
    """path_planner_common DubinsWrapper::fill(path, speed, startTime) (mpc_node.cpp:245-248)."""

    def __init__(self, qi, param, rho, kind, speed, start_time):
        self.qi = tuple(float(v) for v in qi)          # x, y, yaw (yaw: mathematical, 0 = +x, CCW)
        self.param = tuple(float(v) for v in param)    # segment lengths in units of rho
        self.rho, self.kind, self.speed, self.start_time = float(rho), int(kind), float(speed), float(start_time)
        self.end_time = self.start_time + sum(self.param) * self.rho / self.speed   # test_controller.cpp:467

    def pose_at_length(self, arclen):
        tp = arclen / self.rho
        types = _DIRDATA[self.kind]
        qi = (0.0, 0.0, self.qi[2])
        p1, p2 = self.param[0], self.param[1]
        q1 = _segment(p1, qi, types[0])
        q2 = _segment(p2, q1, types[1])
        if tp < p1:
            q = _segment(tp, qi, types[0])
        elif tp < p1 + p2:
            q = _segment(tp - p1, q1, types[1])
        else:
            q = _segment(tp - p1 - p2, q2, types[2])
        return (q[0] * self.rho + self.qi[0], q[1] * self.rho + self.qi[1], _mod2pi(q[2]))

    def end_pose(self):
        return self.pose_at_length(sum(self.param) * self.rho)


class DubinsPlan:
    def __init__(self, wrappers=()):
        self.wrappers = list(wrappers)

    def append(self, w):
        self.wrappers.append(w)

    def empty(self):
        return not self.wrappers

    def start_time(self):
        return self.wrappers[0].start_time

    def end_time(self):
        return self.wrappers[-1].end_time

    def total_time(self):
        return sum(w.end_time - w.start_time for w in self.wrappers)

    def sample(self, t):
        """-> (x, y, heading, speed) with compass heading; times outside the plan clamp to its ends."""
        ws = self.wrappers
        k = None
        for i, w in enumerate(ws):
            if w.start_time <= t <= w.end_time:
                k = i
                break
        tt = t
        if k is None:
            if t < ws[0].start_time:
                k, tt = 0, ws[0].start_time
            else:
                k, tt = len(ws) - 1, ws[-1].end_time
                for i in range(len(ws) - 1):
                    if ws[i].end_time < t < ws[i + 1].start_time:
                        k, tt = i, ws[i].end_time
                        break
        w = ws[k]
        x, y, yaw = w.pose_at_length((tt - w.start_time) * w.speed)
        return (x, y, _mod2pi(math.pi / 2 - yaw), w.speed)

    def checkpoints(self, start_time, depth):
        """controller.cpp:96-98: checkpoint i = plan sampled at start_time + (i + 1) * 1 s."""
        return [self.sample(start_time + (i + 1) * 1.0) for i in range(depth)]

    def max_iterations(self, start_time):
        """controller.cpp:59-61"""
        return int((self.start_time() - start_time + self.total_time()) / 1.0 - 1)


    def to_struct(self):
        """-> binding.Plan (mpc_plan of include/mpc_hip.h) for the device-side sampler"""
        if len(self.wrappers) > B.MAX_PLAN_SEGMENTS:
            raise B.MpcError(8, "plan has more than MPC_MAX_PLAN_SEGMENTS segments")
        pl = B.Plan()
        pl.n_segments = len(self.wrappers)
        for i, w in enumerate(self.wrappers):
            d = pl.segments[i]
            d.qi[:] = w.qi
            d.param[:] = w.param
            d.rho, d.type, d.speed, d.start_time, d.end_time = w.rho, w.kind, w.speed, w.start_time, w.end_time
        return pl


def chain(words, rho, speed, start_pose, start_time):
    """Chain Dubins words [(kind, (p0, p1, p2)), ...], each starting at the previous one's end pose/time."""
    plan = DubinsPlan()
    pose, t = tuple(start_pose), start_time
    for kind, param in words:
        w = DubinsWrapper(pose, param, rho, kind, speed, t)
        plan.append(w)
        pose, t = w.end_pose(), w.end_time
    return plan


# ---- seeded candidate sets (include/mpc_candidates.h) in numpy ---------------------------------------------
_M = np.uint64(0xFFFFFFFFFFFFFFFF)


def _mix64(z):
    with np.errstate(over="ignore"):
        z = z + np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return z ^ (z >> np.uint64(31))


def seeded_controls(seed, first, count, depth, last_rudder=0.0, last_throttle=0.0):
    """-> (rudders, throttles) float64 arrays [depth][count] for ordinals first .. first+count-1."""
    ordinal = np.arange(first, first + count, dtype=np.uint64)
    with np.errstate(over="ignore"):
        base = _mix64(np.uint64(seed) ^ (ordinal * np.uint64(0xD1342543DE82EF95)))
        rud = np.empty((depth, count), dtype=np.float64)
        thr = np.empty((depth, count), dtype=np.float64)
        for l in range(depth):
            h = _mix64(base + np.uint64(l))
            ri = (h & np.uint64(0xFFFFFFFF)) % np.uint64(33)
            ti = (h >> np.uint64(32)) % np.uint64(9)
            rud[l] = -1.0 + 0.0625 * ri.astype(np.float64)
            thr[l] = 0.125 * ti.astype(np.float64)
    is0, is1 = ordinal == 0, ordinal == 1
    rud[:, is0] = 0.0
    thr[:, is0] = 1.0
    rud[:, is1] = last_rudder
    thr[:, is1] = last_throttle
    return rud, thr


# ---- BASELINE configs ------------------------------------------------------------------------------------
SEED = 0xC0FFEE
UNIFORM_STEP = 0.05                      # "0.05 s step": 20 x simulate(r, t, 0.05, current) per 1-s level
T0 = 100.0


def uniform_substeps(step=UNIFORM_STEP):
    n = int(round(1.0 / step))
    return [step] * n


def config2_plan():
    """3 chained Dubins words (LSR, RSL, LRL), rho = 8 m, 2.5 m/s, >= 31 s (serves D = 20 and D = 30)."""
    words = [(LSR, (0.6, 2.0, 0.8)), (RSL, (0.7, 1.8, 0.9)), (LRL, (0.5, 2.4, 0.6))]
    return chain(words, 8.0, 2.5, (0.0, 0.0, math.pi / 2), T0)


def config3_plan():
    """Lawnmower survey: 4 legs of 40 m joined by 180-degree turns of radius 8 m (LSL / RSR, zero middle)."""
    h = math.pi / 2
    words = [(LSL, (0.0, 5.0, 0.0)), (RSR, (h, 0.0, h)), (LSL, (0.0, 5.0, 0.0)), (LSL, (h, 0.0, h)),
             (LSL, (0.0, 5.0, 0.0)), (RSR, (h, 0.0, h)), (LSL, (0.0, 5.0, 0.0))]
    return chain(words, 8.0, 2.5, (0.0, 0.0, math.pi / 2), T0)


def perturbed_start(plan, t0=T0, cross_track=1.0, dheading=0.1, speed=2.0):
    x, y, heading, _ = plan.sample(t0)
    # cross-track offset to starboard of the reference heading
    return (x + cross_track * math.cos(heading), y - cross_track * math.sin(heading),
            _mod2pi(heading + dheading), speed)


def config2_problem(depth=20, current=(0.0, 0.0), plan=None):
    plan = plan or config2_plan()
    return B.make_problem(perturbed_start(plan), plan.checkpoints(T0, depth), uniform_substeps(),
                          last=(0.0, 0.5), current=current)


def config3_problem(depth=20):
    return config2_problem(depth, current=(0.35, -0.20), plan=config3_plan())


def config1_problem(depth=4, length_m=25.0, current=(0.0, 0.0)):
    """test_controller-style plumbing case: straight-line reference (LSL with only the S segment), start on the
    line at 2.5 m/s, reference-native tree search with simulate(r, t, 1.0) levels (11 substeps)."""
    rho = 8.0
    plan = DubinsPlan([DubinsWrapper((0.0, 0.0, math.pi / 2), (0.0, length_m / rho, 0.0), rho, LSL, 2.5, 4.0)])
    start = (0.0, 0.0, 0.0, 2.5)
    depth = min(depth, plan.max_iterations(4.0))
    p = B.make_problem(start, plan.checkpoints(4.0, depth), B.substep_schedule(1.0), last=(0.0, 0.0), current=current)
    return p, plan


def semicircle_problem(length_s, current=(0.2, -0.1), weights=(1.0, 20.0, 5.0)):
    """The reference's own end-to-end scenario (test/test_controller.cpp:449-467): LRL semicircle, rho 8 m, 2.5 m/s, plan
    starting at t = 4, truncated to length_s seconds (the plan's length sets the search depth, controller.cpp:59-61);
    start (0, 0) heading north at 2.5 m/s; reference-native 1-s levels.  -> (problem, depth)"""
    w = DubinsWrapper((0.0, 0.0, math.pi / 2), (0.0, math.pi, 0.0), 8.0, LRL, 2.5, 4.0)
    w.end_time = 4.0 + length_s
    plan = DubinsPlan([w])
    depth = plan.max_iterations(4.0)
    return B.make_problem((0.0, 0.0, 0.0, 2.5), plan.checkpoints(4.0, depth), B.substep_schedule(1.0),
                          last=(0.0, 0.0), current=current, weights=weights), depth


def fleet_problems(n_problems, depth=20, base_seed=SEED, with_plans=False):
    """Config 5: independent vessels; per problem a random Dubins word, start perturbation and current.
    with_plans: also return the DubinsPlan objects (every plan starts at T0)."""
    rng = np.random.RandomState(base_seed & 0x7FFFFFFF)
    probs, seeds, plans = [], [], []
    for i in range(n_problems):
        kind = int(rng.randint(0, 6))
        if kind in (RLR, LRL):
            param = (rng.uniform(0.2, 1.0), rng.uniform(math.pi + 0.2, 1.6 * math.pi), rng.uniform(0.2, 1.0))
        else:
            param = (rng.uniform(0.2, 1.5), rng.uniform(1.0, 4.0), rng.uniform(0.2, 1.5))
        # stretch so that the plan covers depth + 1 seconds
        rho, speed = 8.0, 2.5
        need = (depth + 1.5) * speed / rho
        scale = max(1.0, need / sum(param))
        if kind in (RLR, LRL):
            param = (param[0], param[1], param[2])
            words = [(kind, param), (LSL, (0.0, need, 0.0))]
        else:
            words = [(kind, (param[0], param[1] * scale + need, param[2]))]
        yaw0 = rng.uniform(0, TWO_PI)
        plan = chain(words, rho, speed, (rng.uniform(-500, 500), rng.uniform(-500, 500), yaw0), T0)
        cur = np.clip(rng.normal(0.0, 0.3, size=2), -0.8, 0.8)
        start = perturbed_start(plan, T0, cross_track=rng.uniform(-2, 2), dheading=rng.uniform(-0.3, 0.3),
                                speed=rng.uniform(1.0, 2.75))
        probs.append(B.make_problem(start, plan.checkpoints(T0, depth), uniform_substeps(),
                                    last=(float(rng.randint(-8, 9)) * 0.0625, float(rng.randint(0, 9)) * 0.125),
                                    current=(float(cur[0]), float(cur[1]))))
        seeds.append((base_seed + i) & 0xFFFFFFFFFFFFFFFF)
        plans.append(plan)
    return (probs, seeds, plans) if with_plans else (probs, seeds)
