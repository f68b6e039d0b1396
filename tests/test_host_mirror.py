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


@pytest.mark.gpu
@pytest.mark.parametrize("precision", ["f64", "fast"])
def test_controller_mpc_on_gpu(precision):
    """the same assertions (equality with the oracle's tree search included) whichever model expands the tree"""
    _run("gpu", 110, {"MPC_HIP_TREE_PRECISION": precision})
