"""ctypes binding of the CPU oracle (oracle/mpc_oracle.c).  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package, and only
as the checker.  Nothing under mpc_amd/ imports it (tests/test_layout.py enforces that).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libmpc_oracle.so")


def _up_to_date():
    """libmpc_oracle.so exists and is newer than its sources: no build step, and -- what matters on the GPU box -- no
    child process (a process that has initialised the GPU must not fork + exec: the two-rank worker of
    tests/test_gpu_two_ranks.py died in subprocess.Popen there)"""
    try:
        built = os.path.getmtime(_LIB_PATH)
        return all(os.path.getmtime(os.path.join(_HERE, f)) <= built for f in ("mpc_oracle.c", "mpc_oracle.h", "Makefile", os.path.join("..", "include", "mpc_candidates.h")))
    except OSError:
        return False


def build(force=False):
    if force and os.path.exists(_LIB_PATH):
        os.remove(_LIB_PATH)
    if not _up_to_date():
        subprocess.check_call(["make", "-s", "-C", _HERE, "libmpc_oracle.so"])
    return _LIB_PATH


class State(C.Structure):
    _fields_ = [("x", C.c_double), ("y", C.c_double), ("heading", C.c_double),
                ("speed", C.c_double), ("time", C.c_double)]


class Vehicle(C.Structure):
    _fields_ = [("s", State), ("rpm", C.c_double), ("cmg", C.c_double), ("sog", C.c_double)]


class Config(C.Structure):
    _fields_ = [(n, C.c_double) for n in (
        "rudder_granularity", "throttle_granularity", "w_distance", "w_heading", "w_speed",
        "min_rudder", "mid_rudder", "max_rudder", "min_throttle", "max_throttle", "tolerance",
        "last_rudder", "last_throttle", "current_x", "current_y")]


class Model(C.Structure):
    """orc_model: the boat's 11 parameters (VehicleState.h:86-96)"""
    _fields_ = [(n, C.c_double) for n in (
        "idle_rpm", "max_rpm", "max_rpm_change_rate", "prop_ratio", "prop_pitch", "max_rudder_angle",
        "rudder_coefficient", "rudder_distance", "mass", "max_power", "max_speed")]


class MpcState(C.Structure):
    _fields_ = [("v", Vehicle), ("last_rudder", C.c_double), ("last_throttle", C.c_double),
                ("score", C.c_double)]


class LevelResult(C.Structure):
    _fields_ = [("n_leaves", C.c_uint64), ("best_leaf", C.c_uint64), ("best_score", C.c_double),
                ("best_cumulative", C.c_double)]


class Dubins(C.Structure):
    _fields_ = [("qi", C.c_double * 3), ("param", C.c_double * 3), ("rho", C.c_double), ("type", C.c_int),
                ("speed", C.c_double), ("start_time", C.c_double), ("end_time", C.c_double)]


class Best(C.Structure):
    _fields_ = [("best_index", C.c_uint64), ("best_last", C.c_double), ("best_cumulative", C.c_double)]


class HistoryEntry(C.Structure):
    _fields_ = [("s", State), ("rudder", C.c_double), ("throttle", C.c_double)]


LSL, LSR, RSL, RSR, RLR, LRL = range(6)

_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    try:
        build()                      # no-op when libmpc_oracle.so is newer than its sources
    except Exception:
        if not os.path.exists(_LIB_PATH):
            raise
    L = C.CDLL(_LIB_PATH)
    dp = C.POINTER(C.c_double)
    L.orc_default_config.argtypes = [C.POINTER(Config)]
    L.orc_default_model.argtypes = [C.POINTER(Model)]; L.orc_default_model.restype = None
    L.orc_set_model.argtypes = [C.POINTER(Model)]; L.orc_set_model.restype = None
    L.orc_vehicle_from_state.argtypes = [State]; L.orc_vehicle_from_state.restype = Vehicle
    L.orc_schedule.argtypes = [C.c_double, dp, C.c_int]; L.orc_schedule.restype = C.c_int
    L.orc_estimate.argtypes = [C.POINTER(Vehicle)] + [C.c_double] * 5
    L.orc_simulate.argtypes = [C.POINTER(Vehicle)] + [C.c_double] * 5; L.orc_simulate.restype = Vehicle
    L.orc_heading_difference.argtypes = [C.c_double, C.c_double]; L.orc_heading_difference.restype = C.c_double
    L.orc_compare_states.argtypes = [C.POINTER(State), C.POINTER(Vehicle)] + [C.c_double] * 3
    L.orc_compare_states.restype = C.c_double
    L.orc_dubins_fill.argtypes = [C.POINTER(Dubins), dp, dp, C.c_double, C.c_int, C.c_double, C.c_double]
    L.orc_dubins_sample.argtypes = [C.POINTER(Dubins), C.c_double, dp]
    L.orc_plan_sample.argtypes = [C.POINTER(Dubins), C.c_int, C.POINTER(State)]; L.orc_plan_sample.restype = C.c_int
    L.orc_plan_total_time.argtypes = [C.POINTER(Dubins), C.c_int]; L.orc_plan_total_time.restype = C.c_double
    L.orc_plan_contains_time.argtypes = [C.POINTER(Dubins), C.c_int, C.c_double]
    L.orc_plan_contains_time.restype = C.c_int
    L.orc_max_iterations.argtypes = [C.POINTER(Dubins), C.c_int, C.c_double]; L.orc_max_iterations.restype = C.c_int
    L.orc_mpc_tree.argtypes = [C.POINTER(Config), State, C.POINTER(State), C.c_int, C.c_int64,
                               C.POINTER(MpcState), C.POINTER(LevelResult), C.POINTER(C.c_uint64)]
    L.orc_mpc_tree.restype = C.c_int
    L.orc_eval_candidates.argtypes = [C.POINTER(Config), Vehicle, C.POINTER(State), C.c_int, dp, C.c_int,
                                      dp, dp, C.c_uint64, C.c_int, dp, dp, C.POINTER(Best)]
    L.orc_eval_seeded.argtypes = [C.POINTER(Config), Vehicle, C.POINTER(State), C.c_int, dp, C.c_int,
                                  C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint64, C.c_int, dp, dp,
                                  C.POINTER(Best)]
    L.orc_rollout.argtypes = [C.POINTER(Config), Vehicle, C.POINTER(State), C.c_int, dp, C.c_int, dp, dp,
                              C.POINTER(MpcState)]
    L.orc_seeded_control.argtypes = [C.c_uint64, C.c_uint64, C.c_uint32, C.c_double, C.c_double, dp, dp]
    L.orc_seeded_hash.argtypes = [C.c_uint64, C.c_uint64, C.c_uint32]; L.orc_seeded_hash.restype = C.c_uint64
    L.orc_current_estimate.argtypes =[C.POINTER(HistoryEntry), C.c_int, C.c_int, State, dp, dp]
    _lib = L
    return L


# ---- convenience wrappers ---------------------------------------------------------------------------
def set_model(**kw):
    """process-wide hull of the oracle: the reference's (VehicleState.h:86-96) with the given parameters replaced;
    set_model() with no arguments restores the reference's"""
    m = Model()
    lib().orc_default_model(C.byref(m))
    for k, v in kw.items():
        if not hasattr(m, k):
            raise AttributeError(k)
        setattr(m, k, float(v))
    lib().orc_set_model(C.byref(m))
    return m


def default_config(**kw):
    c = Config()
    lib().orc_default_config(C.byref(c))
    for k, v in kw.items():
        setattr(c, k, v)
    return c


def vehicle(x, y, heading, speed, time=0.0):
    return lib().orc_vehicle_from_state(State(x, y, heading, speed, time))


def simulate(v, rudder, throttle, t, cx=0.0, cy=0.0):
    return lib().orc_simulate(C.byref(v), rudder, throttle, t, cx, cy)


def schedule(t, cap=4096):
    buf = (C.c_double * cap)()
    n = lib().orc_schedule(t, buf, cap)
    return np.array(buf[:n], dtype=np.float64)


def compare_states(ref, sim, wd, wh, ws):
    return lib().orc_compare_states(C.byref(ref), C.byref(sim), wd, wh, ws)


def dubins(qi, param, rho, typ, speed, start_time, end_time=None):
    w = Dubins()
    lib().orc_dubins_fill(C.byref(w), (C.c_double * 3)(*qi), (C.c_double * 3)(*param), rho, typ, speed, start_time)
    if end_time is not None:           # DubinsWrapper::updateEndTime
        w.end_time = end_time
    return w


def plan_array(wrappers):
    return (Dubins * len(wrappers))(*wrappers)


def plan_sample(plan, t):
    s = State(0, 0, 0, 0, t)
    lib().orc_plan_sample(plan, len(plan), C.byref(s))
    return s


def checkpoints(plan, start_time, depth):
    """controller.cpp:96-98: checkpoint i is the plan sampled at start_time + (i + 1)."""
    out = (State * depth)()
    for i in range(depth):
        out[i].time = start_time + (i + 1) * 1.0
        lib().orc_plan_sample(plan, len(plan), C.byref(out[i]))
    return out


def mpc_tree(cfg, start, cps, max_iterations, node_budget=-1):
    traj = (MpcState * (max_iterations + 1))()
    levels = (LevelResult * max(max_iterations, 1))()
    nodes = C.c_uint64(0)
    it = lib().orc_mpc_tree(C.byref(cfg), start, cps, max_iterations, node_budget, traj, levels, C.byref(nodes))
    return it, traj, levels, nodes.value


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double)) if a is not None else None


def eval_candidates(cfg, start_vehicle, cps, depth, substeps, rudders, throttles, cumulative=False):
    rudders = np.ascontiguousarray(rudders, dtype=np.float64)
    throttles = np.ascontiguousarray(throttles, dtype=np.float64)
    assert rudders.shape == throttles.shape and rudders.shape[0] == depth
    n = rudders.shape[1]
    sub = np.ascontiguousarray(substeps, dtype=np.float64)
    last = np.empty(n); cum = np.empty(n); best = Best()
    lib().orc_eval_candidates(C.byref(cfg), start_vehicle, cps, depth, _dp(sub), len(sub), _dp(rudders),
                              _dp(throttles), n, int(cumulative), _dp(last), _dp(cum), C.byref(best))
    return last, cum, best


def eval_seeded(cfg, start_vehicle, cps, depth, substeps, seed, first, stride, count, cumulative=False):
    sub = np.ascontiguousarray(substeps, dtype=np.float64)
    last = np.empty(count); cum = np.empty(count); best = Best()
    lib().orc_eval_seeded(C.byref(cfg), start_vehicle, cps, depth, _dp(sub), len(sub), seed, first, stride,
                          count, int(cumulative), _dp(last), _dp(cum), C.byref(best))
    return last, cum, best


def rollout(cfg, start_vehicle, cps, depth, substeps, rudders, throttles):
    sub = np.ascontiguousarray(substeps, dtype=np.float64)
    r = np.ascontiguousarray(rudders, dtype=np.float64); t = np.ascontiguousarray(throttles, dtype=np.float64)
    traj = (MpcState * (depth + 1))()
    lib().orc_rollout(C.byref(cfg), start_vehicle, cps, depth, _dp(sub), len(sub), _dp(r), _dp(t), traj)
    return traj


def current_estimate(history, enabled, actual):
    arr = (HistoryEntry * max(len(history), 1))(*history)
    cx = C.c_double(); cy = C.c_double()
    lib().orc_current_estimate(arr, len(history), int(enabled), actual, C.byref(cx), C.byref(cy))
    return cx.value, cy.value
