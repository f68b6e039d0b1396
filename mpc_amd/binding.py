"""ctypes binding of libmpc_hip.so (include/mpc_hip.h).

Plumbing only: no arithmetic of the hot path happens here.  Device buffers are passed as integers
(``tensor.data_ptr()``) or as objects with a ``data_ptr()`` method (torch tensors on the handle's GPU).
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libmpc_hip.so")

MAX_DEPTH = 64
MAX_SUBSTEPS = 64
PREC_FAST, PREC_F64 = 0, 1
SELECT_LAST_STEP, SELECT_CUMULATIVE = 0, 1
REFINE_REL = 1.0e-6

# every symbol include/mpc_hip.h declares (tests/test_abi.py checks the built library exports them all)
EXPORTED_SYMBOLS = (
    "mpc_hip_abi_version", "mpc_hip_status_string", "mpc_hip_last_error", "mpc_hip_default_problem",
    "mpc_hip_substep_schedule", "mpc_hip_default_model", "mpc_hip_set_model", "mpc_hip_create", "mpc_hip_destroy", "mpc_hip_synchronize",
    "mpc_hip_eval_candidates", "mpc_hip_eval_seeded", "mpc_hip_eval_seeded_batch",
    "mpc_hip_upload_problems", "mpc_hip_upload_problems_from_plans", "mpc_hip_fetch_checkpoints", "mpc_hip_estimate_currents", "mpc_hip_history_reset", "mpc_hip_history_update", "mpc_hip_estimate_currents_resident", "mpc_hip_run_seeded", "mpc_hip_run_seeded_static", "mpc_hip_run_candidates", "mpc_hip_fetch_results",
    "mpc_hip_device_winners", "mpc_hip_copy_winners", "mpc_hip_set_winner_sink", "mpc_hip_combine_shards", "mpc_hip_combine_shards_exact", "mpc_hip_copy_winner_records", "mpc_hip_set_winner_record_sink", "mpc_hip_set_pipeline", "mpc_hip_flush", "mpc_hip_set_timing", "mpc_hip_timing", "mpc_hip_solve_tree", "mpc_hip_solve_tree_shard", "mpc_hip_solve_tree_deep", "mpc_hip_rollout",
)


class MpcError(RuntimeError):
    def __init__(self, status, text):
        super().__init__(text)
        self.status = status


class Checkpoint(C.Structure):
    _fields_ = [("x", C.c_double), ("y", C.c_double), ("heading", C.c_double), ("speed", C.c_double)]


class LevelState(C.Structure):
    _fields_ = [(n, C.c_double) for n in (
        "x", "y", "heading", "speed", "rpm", "course_made_good", "speed_over_ground",
        "rudder", "throttle", "score")]


class Problem(C.Structure):
    _fields_ = [(n, C.c_double) for n in (
        "start_x", "start_y", "start_heading", "start_speed", "last_rudder", "last_throttle",
        "current_x", "current_y", "w_distance", "w_heading", "w_speed",
        "rudder_granularity", "throttle_granularity", "min_rudder", "mid_rudder", "max_rudder",
        "min_throttle", "max_throttle", "tolerance")] + [
        ("depth", C.c_int32), ("n_substeps", C.c_int32),
        ("checkpoints", Checkpoint * MAX_DEPTH), ("substeps", C.c_double * MAX_SUBSTEPS)]


class Model(C.Structure):
    """mpc_model: the 11 boat parameters of VehicleState.h:86-96"""
    _fields_ = [(n, C.c_double) for n in (
        "idle_rpm", "max_rpm", "max_rpm_change_rate", "prop_ratio", "prop_pitch", "max_rudder_angle",
        "rudder_coefficient", "rudder_distance", "mass", "max_power", "max_speed")]


class Result(C.Structure):
    _fields_ = [("status", C.c_int32), ("levels", C.c_int32), ("best_ordinal", C.c_uint64),
                ("best_last", C.c_double), ("best_cumulative", C.c_double),
                ("first_rudder", C.c_double), ("first_throttle", C.c_double),
                ("evaluated", C.c_uint64), ("rollout_steps", C.c_uint64),
                ("refined", C.c_uint32), ("libm_tie", C.c_uint32),
                ("kernel_ms", C.c_float), ("total_ms", C.c_float),
                ("trajectory", LevelState * (MAX_DEPTH + 1))]


MAX_PLAN_SEGMENTS = 8


class Dubins(C.Structure):
    _fields_ = [("qi", C.c_double * 3), ("param", C.c_double * 3), ("rho", C.c_double),
                ("type", C.c_int32), ("reserved", C.c_int32),
                ("speed", C.c_double), ("start_time", C.c_double), ("end_time", C.c_double)]


class Plan(C.Structure):
    _fields_ = [("n_segments", C.c_int32), ("reserved", C.c_int32), ("segments", Dubins * MAX_PLAN_SEGMENTS)]


CURRENT_MIN_SAMPLES, CURRENT_MAX_SAMPLES = 10, 600


class HistorySample(C.Structure):
    """one CurrentEstimator history entry: the State at an updateEstimate call + the controls in effect"""
    _fields_ = [(n, C.c_double) for n in ("x", "y", "heading", "speed", "time", "rudder", "throttle")]


class TreeLevel(C.Structure):
    _fields_ = [("n_leaves", C.c_uint64), ("best_leaf", C.c_uint64), ("best_score", C.c_double),
                ("best_cumulative", C.c_double), ("libm_tie", C.c_uint32), ("reserved", C.c_uint32)]


TIME_FN = C.CFUNCTYPE(C.c_double, C.c_void_p)

_lib = None


def load_library(path=None):
    """Load libmpc_hip.so.  Raises OSError if it has not been built (python __graft_entry__.py build)."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    p = path or os.environ.get("MPC_HIP_LIB") or LIB_PATH      # MPC_HIP_LIB: another build of the library (A/B measurements)
    # One HIP runtime per process: torch bundles its own libamdhip64.so (same SONAME as /opt/rocm's).  Loading
    # torch FIRST makes libmpc_hip.so bind to that copy; the other order leaves two runtimes in the process and
    # torch then sees no device.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    if not os.path.exists(p):
        raise OSError("%s not found: build it with `make -C mpc_amd/csrc` (there is no CPU fallback)" % p)
    L = C.CDLL(p)
    vp, u64, i32, dp = C.c_void_p, C.c_uint64, C.c_int, C.POINTER(C.c_double)
    PP, RP = C.POINTER(Problem), C.POINTER(Result)
    L.mpc_hip_abi_version.restype = C.c_int
    L.mpc_hip_status_string.argtypes = [i32]; L.mpc_hip_status_string.restype = C.c_char_p
    L.mpc_hip_last_error.restype = C.c_char_p
    L.mpc_hip_default_problem.argtypes = [PP]; L.mpc_hip_default_problem.restype = None
    L.mpc_hip_substep_schedule.argtypes = [C.c_double, dp, i32]; L.mpc_hip_substep_schedule.restype = i32
    L.mpc_hip_default_model.argtypes = [C.POINTER(Model)]; L.mpc_hip_default_model.restype = None
    L.mpc_hip_set_model.argtypes = [vp, C.POINTER(Model), C.c_uint32]
    L.mpc_hip_create.argtypes = [i32, vp, C.POINTER(vp)]
    L.mpc_hip_destroy.argtypes = [vp]; L.mpc_hip_destroy.restype = None
    L.mpc_hip_synchronize.argtypes = [vp]
    L.mpc_hip_eval_candidates.argtypes = [vp, PP, vp, vp, u64, i32, i32, vp, vp, RP]
    L.mpc_hip_eval_seeded.argtypes = [vp, PP, u64, u64, u64, i32, i32, vp, vp, RP]
    L.mpc_hip_eval_seeded_batch.argtypes = [vp, PP, C.POINTER(u64), C.c_uint32, u64, i32, i32, vp, vp, RP]
    L.mpc_hip_upload_problems.argtypes = [vp, PP, C.POINTER(u64), C.c_uint32]
    L.mpc_hip_run_seeded.argtypes = [vp, u64, u64, i32, i32, vp, vp]
    L.mpc_hip_run_seeded_static.argtypes = [vp, u64, u64, i32, i32, vp, vp]
    L.mpc_hip_run_candidates.argtypes = [vp, vp, vp, u64, i32, i32, vp, vp]
    L.mpc_hip_fetch_results.argtypes = [vp, RP, C.c_uint32]
    L.mpc_hip_device_winners.argtypes = [vp, C.POINTER(vp)]
    L.mpc_hip_copy_winners.argtypes = [vp, vp]
    L.mpc_hip_combine_shards.argtypes = [vp, vp, C.c_uint32, C.c_uint32, u64, vp]
    L.mpc_hip_combine_shards_exact.argtypes = [vp, vp, C.c_uint32, C.c_uint32, C.c_uint32, u64, i32, i32, vp, C.POINTER(C.c_uint32)]
    L.mpc_hip_copy_winner_records.argtypes = [vp, vp, C.c_uint32]
    L.mpc_hip_set_winner_record_sink.argtypes = [vp, vp, C.c_uint32]
    L.mpc_hip_upload_problems_from_plans.argtypes = [vp, PP, C.POINTER(u64), C.POINTER(Plan), dp, C.c_uint32]
    L.mpc_hip_fetch_checkpoints.argtypes = [vp, C.c_uint32, C.POINTER(Checkpoint)]
    L.mpc_hip_estimate_currents.argtypes = [vp, C.POINTER(HistorySample), C.POINTER(C.c_uint32), C.c_uint32, dp, C.c_uint32,
                                            i32, dp]
    L.mpc_hip_history_reset.argtypes = [vp, C.c_uint32]
    L.mpc_hip_history_update.argtypes = [vp, C.POINTER(HistorySample), C.c_uint32]
    L.mpc_hip_estimate_currents_resident.argtypes = [vp, dp, C.c_uint32, i32, dp]
    L.mpc_hip_set_winner_sink.argtypes = [vp, vp]
    L.mpc_hip_set_pipeline.argtypes = [vp, i32]
    L.mpc_hip_flush.argtypes = [vp]
    L.mpc_hip_set_timing.argtypes = [vp, C.c_uint32]
    L.mpc_hip_timing.argtypes = [vp, i32, C.POINTER(u64), dp, dp]
    L.mpc_hip_solve_tree.argtypes = [vp, PP, i32, i32, TIME_FN, vp, C.c_double, u64, C.POINTER(TreeLevel), RP]
    L.mpc_hip_solve_tree_shard.argtypes = [vp, PP, i32, i32, TIME_FN, vp, C.c_double, u64, i32, C.c_uint32, C.c_uint32,
                                           C.POINTER(TreeLevel), RP]
    L.mpc_hip_solve_tree_deep.argtypes = [vp, PP, i32, i32, TIME_FN, vp, C.c_double, C.POINTER(TreeLevel), RP]
    L.mpc_hip_rollout.argtypes = [vp, PP, dp, dp, i32, RP]
    if path is None:
        _lib = L
    return L


def _check(status):
    if status != 0:
        L = load_library()
        detail = L.mpc_hip_last_error().decode()
        text = L.mpc_hip_status_string(status).decode()
        raise MpcError(status, "%s%s" % (text, (": " + detail) if detail else ""))


def _ptr(x):
    if x is None:
        return None
    if hasattr(x, "data_ptr"):
        return x.data_ptr()
    return int(x)


def default_problem():
    p = Problem()
    load_library().mpc_hip_default_problem(C.byref(p))
    return p


def default_model(**kw):
    """the reference's hull (VehicleState.h:86-96), with any of the 11 parameters replaced by keyword"""
    m = Model()
    load_library().mpc_hip_default_model(C.byref(m))
    for k, v in kw.items():
        if not hasattr(m, k):
            raise AttributeError(k)
        setattr(m, k, float(v))
    return m


def substep_schedule(t, cap=4096):
    buf = (C.c_double * cap)()
    n = load_library().mpc_hip_substep_schedule(float(t), buf, cap)
    return np.array(buf[:min(n, cap)], dtype=np.float64)


def make_problem(start, checkpoints, substeps, last=(0.0, 0.0), current=(0.0, 0.0), weights=None, **kw):
    """start = (x, y, heading, speed); checkpoints = iterable of (x, y, heading, speed);
    substeps = iterable of substep lengths (one level).  Other fields keep the reference defaults."""
    p = default_problem()
    p.start_x, p.start_y, p.start_heading, p.start_speed = [float(v) for v in start]
    p.last_rudder, p.last_throttle = [float(v) for v in last]
    p.current_x, p.current_y = [float(v) for v in current]
    if weights is not None:
        p.w_distance, p.w_heading, p.w_speed = [float(v) for v in weights]
    cps = list(checkpoints)
    sub = list(substeps)
    if len(cps) > MAX_DEPTH or len(sub) > MAX_SUBSTEPS:
        raise MpcError(8, "problem exceeds MPC_MAX_DEPTH / MPC_MAX_SUBSTEPS")
    p.depth = len(cps)
    p.n_substeps = len(sub)
    for i, c in enumerate(cps):
        p.checkpoints[i] = Checkpoint(*[float(v) for v in c])
    for i, s in enumerate(sub):
        p.substeps[i] = float(s)
    for k, v in kw.items():
        if not hasattr(p, k):
            raise AttributeError(k)
        setattr(p, k, v)
    return p


def winner_record_words(depth):
    """MPC_WINNER_RECORD_WORDS: (score bits, ordinal) + (rudder, throttle) per level"""
    return 2 + 2 * int(depth)


class Solver:
    """One mpc_hip_handle.  ``stream`` is an integer hipStream_t (e.g. torch.cuda.current_stream().cuda_stream)
    or None for a stream owned by the handle."""

    def __init__(self, device=0, stream=None):
        self._L = load_library()
        h = C.c_void_p()
        _check(self._L.mpc_hip_create(int(device), C.c_void_p(stream) if stream else None, C.byref(h)))
        self._h = h
        self.device = int(device)
        self._n_problems = 0

    def close(self):
        if getattr(self, "_h", None):
            self._L.mpc_hip_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def synchronize(self):
        _check(self._L.mpc_hip_synchronize(self._h))

    def set_model(self, models):
        """mpc_hip_set_model: one Model (every problem) or a list (problem p uses models[p % len])"""
        ms = list(models) if isinstance(models, (list, tuple)) else [models]
        arr = (Model * len(ms))(*ms)
        _check(self._L.mpc_hip_set_model(self._h, arr, len(ms)))
        self._n_problems = 0

    # ---- one-call forms ------------------------------------------------------------------------------
    def eval_candidates(self, problem, d_rudders, d_throttles, n, precision=PREC_FAST,
                        selection=SELECT_LAST_STEP, d_cost_last=None, d_cost_cum=None):
        r = Result()
        _check(self._L.mpc_hip_eval_candidates(self._h, C.byref(problem), _ptr(d_rudders), _ptr(d_throttles), n,
                                               precision, selection, _ptr(d_cost_last), _ptr(d_cost_cum), C.byref(r)))
        return r

    def eval_seeded(self, problem, seed, first, count, precision=PREC_FAST, selection=SELECT_LAST_STEP,
                    d_cost_last=None, d_cost_cum=None):
        r = Result()
        _check(self._L.mpc_hip_eval_seeded(self._h, C.byref(problem), seed, first, count, precision, selection,
                                           _ptr(d_cost_last), _ptr(d_cost_cum), C.byref(r)))
        return r

    def eval_seeded_batch(self, problems, seeds, count, precision=PREC_FAST, selection=SELECT_LAST_STEP,
                          d_cost_last=None, d_cost_cum=None):
        n = len(problems)
        arr = (Problem * n)(*problems)
        sd = (C.c_uint64 * n)(*seeds)
        res = (Result * n)()
        _check(self._L.mpc_hip_eval_seeded_batch(self._h, arr, sd, n, count, precision, selection,
                                                 _ptr(d_cost_last), _ptr(d_cost_cum), res))
        return res

    # ---- split forms -----------------------------------------------------------------------------------
    def upload(self, problems, seeds=None):
        n = len(problems)
        arr = (Problem * n)(*problems)
        sd = (C.c_uint64 * n)(*seeds) if seeds is not None else None
        _check(self._L.mpc_hip_upload_problems(self._h, arr, sd, n))
        self._n_problems = n

    def upload_from_plans(self, problems, seeds, plans, start_times):
        """upload(), with every problem's checkpoints sampled on the device from its Dubins plan (binding.Plan) at
        start_times[p] + 1, 2, ... s; the problems' own checkpoints are ignored."""
        n = len(problems)
        arr = (Problem * n)(*problems)
        sd = (C.c_uint64 * n)(*seeds) if seeds is not None else None
        pl = (Plan * n)(*plans)
        t0 = (C.c_double * n)(*[float(t) for t in start_times])
        _check(self._L.mpc_hip_upload_problems_from_plans(self._h, arr, sd, pl, t0, n))
        self._n_problems = n

    def fetch_checkpoints(self, problem, depth):
        out = (Checkpoint * depth)()
        _check(self._L.mpc_hip_fetch_checkpoints(self._h, int(problem), out))
        return [(c.x, c.y, c.heading, c.speed) for c in out]

    def estimate_currents(self, histories, actual, apply=False):
        """Batched CurrentEstimator::getCurrent on the device.  histories: per vessel a list of HistorySample (oldest
        first) or of 7-tuples (x, y, heading, speed, time, rudder, throttle); actual: per vessel (x, y, time).
        apply: the estimates become the resident problems' currents.  -> [(cx, cy)] per vessel."""
        n = len(histories)
        stride = max(1, max(len(hh) for hh in histories))
        flat = (HistorySample * (n * stride))()
        for p, hh in enumerate(histories):
            for i, e in enumerate(hh):
                flat[p * stride + i] = e if isinstance(e, HistorySample) else HistorySample(*[float(v) for v in e])
        counts = (C.c_uint32 * n)(*[len(hh) for hh in histories])
        act = (C.c_double * (3 * n))(*[float(v) for a3 in actual for v in a3])
        out = (C.c_double * (2 * n))()
        _check(self._L.mpc_hip_estimate_currents(self._h, flat, counts, stride, act, n, int(bool(apply)), out))
        return [(out[2 * p], out[2 * p + 1]) for p in range(n)]

    def history_reset(self, n_vessels):
        """device-resident CurrentEstimator state for a fleet: empty rings, 20 skips left (resetEstimate)"""
        _check(self._L.mpc_hip_history_reset(self._h, int(n_vessels)))

    def history_update(self, samples):
        """updateEstimate for every vessel: samples[p] = HistorySample or 7-tuple (x, y, heading, speed, time, rudder, throttle)"""
        n = len(samples)
        arr = samples if isinstance(samples, C.Array) else (HistorySample * n)(
            *[e if isinstance(e, HistorySample) else HistorySample(*[float(v) for v in e]) for e in samples])
        _check(self._L.mpc_hip_history_update(self._h, arr, n))

    def estimate_currents_resident(self, actual, apply=False):
        """getCurrent for every vessel from the device-resident rings; actual: per vessel (x, y, time) -> [(cx, cy)]"""
        n = len(actual)
        act = (C.c_double * (3 * n))(*[float(v) for a3 in actual for v in a3])
        out = (C.c_double * (2 * n))()
        _check(self._L.mpc_hip_estimate_currents_resident(self._h, act, n, int(bool(apply)), out))
        return [(out[2 * p], out[2 * p + 1]) for p in range(n)]

    def run_seeded(self, first, count, precision=PREC_FAST, selection=SELECT_LAST_STEP, d_cost_last=None,
                   d_cost_cum=None):
        _check(self._L.mpc_hip_run_seeded(self._h, first, count, precision, selection, _ptr(d_cost_last),
                                          _ptr(d_cost_cum)))

    def run_seeded_static(self, first, count, precision=PREC_FAST, selection=SELECT_LAST_STEP, d_cost_last=None,
                          d_cost_cum=None):
        """run_seeded with the library's fixed device array as working memory: allocates nothing (mpc_hip.h)"""
        _check(self._L.mpc_hip_run_seeded_static(self._h, first, count, precision, selection, _ptr(d_cost_last),
                                                 _ptr(d_cost_cum)))

    def run_candidates(self, d_rudders, d_throttles, n, precision=PREC_FAST, selection=SELECT_LAST_STEP,
                       d_cost_last=None, d_cost_cum=None):
        _check(self._L.mpc_hip_run_candidates(self._h, _ptr(d_rudders), _ptr(d_throttles), n, precision, selection,
                                              _ptr(d_cost_last), _ptr(d_cost_cum)))

    def fetch(self):
        res = (Result * self._n_problems)()
        _check(self._L.mpc_hip_fetch_results(self._h, res, self._n_problems))
        return res

    def device_winners(self):
        p = C.c_void_p()
        _check(self._L.mpc_hip_device_winners(self._h, C.byref(p)))
        return p.value

    def copy_winners(self, d_dst):
        _check(self._L.mpc_hip_copy_winners(self._h, _ptr(d_dst)))

    def combine_shards(self, d_gathered, n_shards, n_problems, ordinal_stride, d_out):
        _check(self._L.mpc_hip_combine_shards(self._h, _ptr(d_gathered), n_shards, n_problems, ordinal_stride, _ptr(d_out)))

    def combine_shards_exact(self, d_gathered, n_shards, n_rows, depth, ordinal_stride, d_out, precision=PREC_FAST,
                             selection=SELECT_LAST_STEP):
        """the exact cross-shard decision on gathered winner records of winner_record_words(depth) words (mpc_hip.h);
        precision / selection: those of the runs that produced the records"""
        _check(self._L.mpc_hip_combine_shards_exact(self._h, _ptr(d_gathered), n_shards, n_rows, int(depth), ordinal_stride,
                                                    int(precision), int(selection), _ptr(d_out), None))

    def tied_rows(self):
        """rows that had a cross-shard near-tie in this handle's combine_shards_exact calls so far; synchronises"""
        n = C.c_uint32(0)
        _check(self._L.mpc_hip_combine_shards_exact(self._h, None, 0, 0, 0, 0, 0, 0, None, C.byref(n)))
        return n.value

    def copy_winner_records(self, d_dst, depth):
        """the last run's winners with their controls: winner_record_words(depth) int64 per problem"""
        _check(self._L.mpc_hip_copy_winner_records(self._h, _ptr(d_dst), int(depth)))

    def set_winner_record_sink(self, d_dst, depth):
        """runs also write their winner records (with controls) to d_dst (None: off)"""
        _check(self._L.mpc_hip_set_winner_record_sink(self._h, _ptr(d_dst), int(depth)))

    def set_winner_sink(self, d_dst):
        """runs also write their packed winners to d_dst (a device tensor / pointer; None: off)"""
        _check(self._L.mpc_hip_set_winner_sink(self._h, _ptr(d_dst)))

    def set_pipeline(self, on):
        """deferred finish for streams of solves (mpc_hip.h): run n's results appear with run n+1 or on demand"""
        _check(self._L.mpc_hip_set_pipeline(self._h, int(on)))      # False/0 off, True/1 deferred finish, K = 2..4: K runs per launch

    def flush(self):
        """launch a pending finish now (before a collective that reads the winner sink on the stream)"""
        _check(self._L.mpc_hip_flush(self._h))

    def set_timing(self, period):
        """HIP events on every `period`-th run (1 = all, 0 = none)."""
        _check(self._L.mpc_hip_set_timing(self._h, int(period)))

    def timing(self, reset=True):
        """-> (timed runs, kernel_ms_sum, total_ms_sum) over the timed runs since the last reset; synchronises."""
        n, k, t = C.c_uint64(), C.c_double(), C.c_double()
        _check(self._L.mpc_hip_timing(self._h, int(reset), C.byref(n), C.byref(k), C.byref(t)))
        return n.value, k.value, t.value

    # ---- tree search / single rollout --------------------------------------------------------------------
    def solve_tree(self, problem, max_levels, precision=PREC_F64, now=None, end_time=0.0, max_nodes=0,
                   split_level=0, shard=0, n_shards=1):
        """n_shards > 1: this GPU's share of a search split after `split_level` (mpc_hip_solve_tree_shard); combine
        the shards' levels with mpc_amd.distributed.combine_tree_levels."""
        r = Result()
        levels = (TreeLevel * max(max_levels, 1))()
        cb = TIME_FN(lambda ctx: float(now())) if now is not None else C.cast(None, TIME_FN)
        if n_shards == 1:
            _check(self._L.mpc_hip_solve_tree(self._h, C.byref(problem), max_levels, precision, cb, None,
                                              float(end_time), max_nodes, levels, C.byref(r)))
        else:
            _check(self._L.mpc_hip_solve_tree_shard(self._h, C.byref(problem), max_levels, precision, cb, None,
                                                    float(end_time), max_nodes, split_level, shard, n_shards,
                                                    levels, C.byref(r)))
        return r, levels

    def solve_tree_deep(self, problem, max_levels, precision=PREC_F64, now=None, end_time=0.0):
        """mpc_hip_solve_tree_deep: the plain search, then level by level in pieces beyond what one expansion holds"""
        r = Result()
        levels = (TreeLevel * max(max_levels, 1))()
        cb = TIME_FN(lambda ctx: float(now())) if now is not None else C.cast(None, TIME_FN)
        _check(self._L.mpc_hip_solve_tree_deep(self._h, C.byref(problem), max_levels, precision, cb, None, float(end_time),
                                               levels, C.byref(r)))
        return r, levels

    def rollout(self, problem, rudders, throttles, precision=PREC_F64):
        r = Result()
        n = problem.depth
        ru = (C.c_double * n)(*[float(v) for v in rudders])
        th = (C.c_double * n)(*[float(v) for v in throttles])
        _check(self._L.mpc_hip_rollout(self._h, C.byref(problem), ru, th, precision, C.byref(r)))
        return r
