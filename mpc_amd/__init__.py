"""mpc_amd -- MI355X-native implementation of the afb2001/mpc trajectory-optimisation hot path.

The product is the C-ABI library ``libmpc_hip.so`` (include/mpc_hip.h; sources in mpc_amd/csrc).  This
package is the thin Python plumbing over it used by the tests and bench.py: a ctypes binding, helpers that
build ``mpc_problem`` records, and the multi-GPU combine step over torch.distributed (RCCL).  There is no
CPU fallback: if the library is missing or no HIP device is present, calls fail.
"""
from .binding import (  # noqa: F401
    LIB_PATH, MpcError, Problem, Result, LevelState, Checkpoint, Model, default_model, TreeLevel, Dubins, Plan, HistorySample, Solver, winner_record_words,
    PREC_FAST, PREC_F64, SELECT_LAST_STEP, SELECT_CUMULATIVE, MAX_DEPTH, MAX_SUBSTEPS, REFINE_REL,
    load_library, default_problem, substep_schedule, make_problem, EXPORTED_SYMBOLS,
)
