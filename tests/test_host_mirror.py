"""The C++ host-side mirror of the reference's Controller API (mpc_amd/host) through its gtest-free runner, whose
cases follow test/test_controller.cpp.  `cpu` needs no device; `gpu` drives Controller::mpc through the C-ABI."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "tests", "host", "host_tests")


def _run(mode, timeout, env=None):
    assert os.path.exists(EXE), "build it: python __graft_entry__.py (make -C tests/host)"
    p = subprocess.run([EXE, mode], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=timeout, text=True,
                       env=dict(os.environ, **(env or {})))
    print(p.stdout[-3000:])
    assert p.returncode == 0, p.stdout[-3000:]
    assert "0 failed" in p.stdout


def test_host_classes_cpu():
    _run("cpu", 120)


REFERENCE_FLAGS = ["-std=c++11", "-pthread", "-funsafe-math-optimizations"]       # the reference's CMakeLists.txt:4-6


def test_drop_in_compiles_with_the_references_own_flags(tmp_path):
    """controller_mpc_hip.cpp is added to the reference's `controller` library, which is built with
    add_compile_options(-std=c++11) and -pthread -funsafe-math-optimizations: every host source compiles cleanly under
    exactly those options, both Makefiles build with them, and the runner built from them is what the tests above and
    below execute (its CPU cases compare the host VehicleState with the oracle bit for bit UNDER these flags)."""
    host = os.path.join(ROOT, "mpc_amd", "host")
    for src in ("controller_mpc_hip.cpp", "controller.cpp", "CurrentEstimator.cpp"):
        subprocess.check_call(["g++"] + REFERENCE_FLAGS + ["-Wall", "-Wextra", "-Werror", "-fsyntax-only",
                                                           "-I" + os.path.join(host, "compat"), "-I" + host,
                                                           "-I" + os.path.join(ROOT, "include"), os.path.join(host, src)])
    for mk in (os.path.join(host, "Makefile"), os.path.join(ROOT, "tests", "host", "Makefile")):
        text = open(mk).read()
        assert "REFERENCE_FLAGS = " + " ".join(REFERENCE_FLAGS) in text and "c++14" not in text and "c++17" not in text, mk
    # a fresh build of the runner from these flags, independent of the in-tree binary
    exe = tmp_path / "host_tests_cxx11"
    oracle_o = tmp_path / "mpc_oracle.o"
    subprocess.check_call(["gcc", "-O2", "-std=c11", "-fPIC", "-ffp-contract=off", "-fno-fast-math", "-D_GNU_SOURCE", "-c",
                           os.path.join(ROOT, "oracle", "mpc_oracle.c"), "-o", str(oracle_o)])
    subprocess.check_call(["g++"] + REFERENCE_FLAGS + ["-O2", "-I" + os.path.join(host, "compat"), "-I" + host, "-I" + os.path.join(ROOT, "include"), "-o", str(exe),
                           os.path.join(ROOT, "tests", "host", "host_tests.cpp")] +
                          [os.path.join(host, f) for f in ("controller.cpp", "controller_mpc_hip.cpp", "CurrentEstimator.cpp")] +
                          [str(oracle_o), "-L" + os.path.join(ROOT, "mpc_amd"), "-lmpc_hip", "-lm",
                           "-Wl,-rpath," + os.path.join(ROOT, "mpc_amd")])
    p = subprocess.run([str(exe), "cpu"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=120, text=True)
    assert p.returncode == 0 and "0 failed" in p.stdout, p.stdout[-2000:]


def test_drop_in_compiles_against_the_references_own_header():
    """tools/check_drop_in.py lays controller_mpc_hip.cpp beside copies of the reference's controller.h,
    control_receiver.h, VehicleState.h and CurrentEstimator.h (as INTEGRATION.md section 1 has a maintainer do) and
    compiles it with the reference's options under -Werror.  Where the reference tree does not exist (the GPU box) the
    tool says so (status 77) and the test is skipped."""
    import sys
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "check_drop_in.py")], stdout=subprocess.PIPE,
                       stderr=subprocess.STDOUT, text=True, timeout=120)
    if p.returncode == 77:
        pytest.skip(p.stdout.strip())
    assert p.returncode == 0, p.stdout[-3000:]
    assert "defines Controller::mpc only" in p.stdout


def test_two_overlapping_solves_under_thread_sanitizer():
    """SURVEY.md section 7 hard part 7 (controller.cpp:267 and :385 overlap, meeting at :295): the runner's two-thread
    cases -- two Controller::mpc calls at once on one Controller, and a solve running while the trajectory number moves
    on (:30-32, :83-86) -- built with -fsanitize=thread from the host sources themselves against a TEST DOUBLE of the
    C-ABI (tests/host/tsan_double_mpc_hip.cpp; an instrumented process cannot host the HIP runtime, and the double aborts
    if a handle is ever inside two calls at once).  The same cases run against the real library on the GPU below."""
    exe = os.path.join(ROOT, "tests", "host", "host_tests_tsan")
    assert os.path.exists(exe), "build it: python __graft_entry__.py (make -C tests/host)"
    nm = subprocess.run(["nm", exe], stdout=subprocess.PIPE, text=True).stdout
    assert "__tsan_init" in nm, "host_tests_tsan is not instrumented"
    p = subprocess.run([exe, "threads"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=300, text=True,
                       env=dict(os.environ, TSAN_OPTIONS="halt_on_error=1 exitcode=66"))
    print(p.stdout[-3000:])
    assert "ThreadSanitizer" not in p.stdout, p.stdout[-3000:]
    assert p.returncode == 0 and "0 failed" in p.stdout, p.stdout[-3000:]


def test_host_library_exports_what_its_header_declares():
    """include/controller_mpc_hip.h: the drop-in's one extra C entry point (the hull) is exported by libcontroller_hip.so"""
    import re
    text = re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "include", "controller_mpc_hip.h")).read(), flags=re.S)
    declared = sorted(set(re.findall(r"\b(mpc_controller_\w+)\s*\(", text)))
    assert declared == ["mpc_controller_set_hull"]
    nm = subprocess.run(["nm", "-D", "--defined-only", os.path.join(ROOT, "mpc_amd", "host", "libcontroller_hip.so")],
                        stdout=subprocess.PIPE, text=True).stdout
    for name in declared:
        assert re.search(r" T %s$" % name, nm, flags=re.M), name


@pytest.mark.gpu
@pytest.mark.parametrize("precision", ["f64", "fast", "auto"])
def test_controller_mpc_on_gpu(precision):
    """the same assertions (equality with the oracle's tree search included) whichever model expands the tree; auto =
    the drop-in's default: the plan's length chooses"""
    _run("gpu", 110, {"MPC_HIP_TREE_PRECISION": precision})
