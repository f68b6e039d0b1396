"""The C-ABI library loads on a machine without a GPU, exports every symbol include/mpc_hip.h declares, agrees with
the header on struct layout, and refuses to run without a device (no CPU fallback).  CPU only, no compute calls."""
import ctypes as C
import os
import re
import subprocess
import sys

import numpy as np
import pytest

import mpc_amd as M
import oracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "mpc_hip.h")


def _declared_functions():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(mpc_hip_\w+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    lib = M.load_library()
    declared = _declared_functions()
    assert len(declared) >= 18
    for name in declared:
        assert hasattr(lib, name), name
    assert sorted(M.EXPORTED_SYMBOLS) == declared          # the Python binding covers the whole header
    assert lib.mpc_hip_abi_version() == 2


def test_struct_layout_matches_the_header(tmp_path):
    src = tmp_path / "sizes.c"
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "mpc_hip.h"\nint main(void){printf("%zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu\\n",'
                   'sizeof(mpc_problem),sizeof(mpc_result),sizeof(mpc_level_state),sizeof(mpc_checkpoint),'
                   'offsetof(mpc_problem,depth),offsetof(mpc_problem,substeps),offsetof(mpc_result,trajectory),sizeof(mpc_tree_level),'
                   'sizeof(mpc_dubins),sizeof(mpc_plan),offsetof(mpc_dubins,speed));return 0;}\n')
    exe = tmp_path / "sizes"
    subprocess.check_call(["gcc", "-std=c11", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)])
    got = [int(v) for v in subprocess.check_output([str(exe)]).split()]
    want = [C.sizeof(M.Problem), C.sizeof(M.Result), C.sizeof(M.LevelState), C.sizeof(M.Checkpoint),
            M.Problem.depth.offset, M.Problem.substeps.offset, M.Result.trajectory.offset, C.sizeof(M.TreeLevel),
            C.sizeof(M.Dubins), C.sizeof(M.Plan), M.Dubins.speed.offset]
    assert got == want


def test_header_is_plain_c_and_constants_agree():
    text = open(HEADER).read()
    code = re.sub(r"/\*.*?\*/", "", text, flags=re.S)                 # declarations only: no C++ / torch types
    assert 'extern "C"' in text and "torch" not in code and "std::" not in code and "class " not in code
    assert int(re.search(r"#define MPC_MAX_DEPTH (\d+)", text).group(1)) == M.MAX_DEPTH
    assert int(re.search(r"#define MPC_MAX_SUBSTEPS (\d+)", text).group(1)) == M.MAX_SUBSTEPS
    assert int(re.search(r"#define MPC_MAX_PLAN_SEGMENTS (\d+)", text).group(1)) == M.binding.MAX_PLAN_SEGMENTS
    assert float(re.search(r"#define MPC_REFINE_REL ([0-9.eE+-]+)", text).group(1)) == M.REFINE_REL
    assert int(re.search(r"#define MPC_CURRENT_MIN_SAMPLES (\d+)", text).group(1)) == M.binding.CURRENT_MIN_SAMPLES == 10
    assert int(re.search(r"#define MPC_CURRENT_MAX_SAMPLES (\d+)", text).group(1)) == M.binding.CURRENT_MAX_SAMPLES == 600
    assert C.sizeof(M.HistorySample) == 7 * 8            # mpc_history_sample: seven doubles, no padding


def test_defaults_are_the_references():
    p = M.default_problem()
    assert (p.rudder_granularity, p.throttle_granularity) == (0.0625, 0.125)          # cfg/mpc.cfg:8-9
    assert (p.w_distance, p.w_heading, p.w_speed) == (1.0, 20.0, 5.0)                # cfg/mpc.cfg:11-15
    assert (p.min_rudder, p.mid_rudder, p.max_rudder, p.min_throttle, p.max_throttle) == (-1, 0, 1, 0, 1)
    assert p.tolerance == 1e-5


def test_substep_schedule_equals_the_oracles():
    for t in (1.0, 0.05, 0.1, 0.75, 2.0, 5.0, 0.30000000000000004):
        assert np.array_equal(M.substep_schedule(t), O.schedule(t)), t


def test_status_strings_carry_the_reference_exception_texts():
    lib = M.load_library()
    assert lib.mpc_hip_status_string(4) == b"Cannot run MPC on an empty plan"                 # controller.cpp:37
    assert lib.mpc_hip_status_string(5) == b"MPC failed to complete a single iteration"       # controller.cpp:196
    assert lib.mpc_hip_status_string(6) == b"MPC selected non-finite controls"                # controller.cpp:202


def test_no_device_means_an_error_not_a_fallback():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(M.MpcError) as e:
        M.Solver(0)
    assert e.value.status == 2
