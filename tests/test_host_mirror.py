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
                                                           os.path.join(host, src)])
    for mk in (os.path.join(host, "Makefile"), os.path.join(ROOT, "tests", "host", "Makefile")):
        text = open(mk).read()
        assert "REFERENCE_FLAGS = " + " ".join(REFERENCE_FLAGS) in text and "c++14" not in text and "c++17" not in text, mk
    # a fresh build of the runner from these flags, independent of the in-tree binary
    exe = tmp_path / "host_tests_cxx11"
    oracle_o = tmp_path / "mpc_oracle.o"
    subprocess.check_call(["gcc", "-O2", "-std=c11", "-fPIC", "-ffp-contract=off", "-fno-fast-math", "-D_GNU_SOURCE", "-c",
                           os.path.join(ROOT, "oracle", "mpc_oracle.c"), "-o", str(oracle_o)])
    subprocess.check_call(["g++"] + REFERENCE_FLAGS + ["-O2", "-I" + os.path.join(host, "compat"), "-I" + host, "-o", str(exe),
                           os.path.join(ROOT, "tests", "host", "host_tests.cpp")] +
                          [os.path.join(host, f) for f in ("controller.cpp", "controller_mpc_hip.cpp", "CurrentEstimator.cpp")] +
                          [str(oracle_o), "-L" + os.path.join(ROOT, "mpc_amd"), "-lmpc_hip", "-lm",
                           "-Wl,-rpath," + os.path.join(ROOT, "mpc_amd")])
    p = subprocess.run([str(exe), "cpu"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=120, text=True)
    assert p.returncode == 0 and "0 failed" in p.stdout, p.stdout[-2000:]


@pytest.mark.gpu
@pytest.mark.parametrize("precision", ["f64", "fast", "auto"])
def test_controller_mpc_on_gpu(precision):
    """the same assertions (equality with the oracle's tree search included) whichever model expands the tree; auto =
    the drop-in's default: the plan's length chooses"""
    _run("gpu", 110, {"MPC_HIP_TREE_PRECISION": precision})
