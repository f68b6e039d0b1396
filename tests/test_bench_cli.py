"""bench.py's launcher contract, checked without a GPU: `--gpus N` run WITHOUT a launcher starts the N ranks itself
(torch.distributed.run on 127.0.0.1, before anything touches the GPU) and relays their exit status; every rank then
refuses to run on a CPU (there is no CPU fallback for the measured path).  CPU only."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _gpu_present():
    import torch
    return torch.cuda.is_available()


@pytest.mark.skipif(_gpu_present(), reason="a GPU is present: the ranks would run the benchmark")
def test_gpus_n_without_a_launcher_starts_the_ranks_itself():
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300, env=env)
    assert p.returncode != 0                                        # relayed from the ranks
    assert "bench.py needs a GPU" in p.stderr                       # a rank got as far as the device check
    assert "launch with torch.distributed.run" not in p.stderr      # the old refusal is gone
    assert p.stdout.strip() == ""                                   # nothing but a JSON line ever goes to stdout


@pytest.mark.skipif(_gpu_present(), reason="a GPU is present")
def test_one_gpu_refuses_the_cpu():
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "1", "--warmup", "0"],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)
    assert p.returncode != 0 and "needs a GPU" in p.stderr and p.stdout.strip() == ""
