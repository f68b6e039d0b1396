"""bench.py's output contract on the GPU: ONE JSON line on stdout carrying the driver's keys, the roofline and (at N=1)
the CPU baseline; the deferred finish and the two-kernel form choose the same control; the RCCL combine rehearsed with
one rank reports its ranks."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench(*args):
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + list(args), stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                       text=True, timeout=280)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, p.stdout[-2000:]
    return json.loads(lines[0])


def test_default_line_has_the_contract_keys():
    d = _bench("--steps", "40", "--warmup", "5")
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline", "timed_region_ms"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 40 and d["warmup"] == 5 and d["unit"] == "steps/s" and d["higher_is_better"] is True
    assert d["vs_baseline"] is None and d["dtype"] == "f64" and d["data"] == "synthetic" and "workload" in d["config"]
    assert abs(d["value"] - 65536 * 400 * 40 / (d["timed_region_ms"] * 1e-3)) <= 1e-6 * d["value"]
    r = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in r, k
    assert r["unit"] == "TFLOP/s" and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12 and 0.2 < r["frac"] < 1.0
    assert r["traffic"] is None or r["traffic"] > 1e6
    c = d["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in c, k
    assert c["kind"] == "port" and c["cores"] == 1 and c["same_choice_as_gpu"] is True and c["value"] > 1e6
    assert d["value"] > 1e11                                       # north_star's floor on one MI355X
    assert d["p50_solve_latency_ms"] < 5.0                         # ... and its latency bound
    assert d["config"]["deferred_finish"] is True
    # the default stream of config-2 solves: four per rollout launch, and the roofline counts a launch as four solves
    assert d["config"]["pipeline_level"] == 4 and d["config"]["solves_per_launch"] == 4
    assert r["kernel"] == "rollout_group_kernel" and r["solves_per_launch"] == 4
    assert r["algorithmic_flop_per_launch"] == 4 * 65536 * 20 * (54 * 20 + 21)
    # the kernel's time (HIP events on the dispatches) accounts for the region: 10 launches of four solves
    assert 0.85 * d["ms_per_step"] <= d["rollout_kernel_ms"] / 4 <= 1.05 * d["ms_per_step"]
    # a stream of DISTINCT solves: four candidate sets in turn, the CPU baseline checks the GPU's choice for every one
    assert d["config"]["distinct_candidate_sets"] == 4 and len(d["chosen"]["per_set"]) == 4
    assert len({c["ordinal"] for c in d["chosen"]["per_set"]}) == 4
    assert c["same_choice_per_set"] == [True] * 4
    assert d["chosen"]["ordinal"] == d["chosen"]["per_set"][(40 - 1) % 4]["ordinal"]
    # the production path (the reference-native tree search through the C-ABI) is in the line
    t = d["tree"]
    for name in ("lrl_semicircle", "config1_straight_line"):
        sc = t["scenarios"][name]
        for depth in ("3", "4", "5", "7", "8"):
            row = sc["depth"][depth]
            assert row["strict_and_fast_agree"] is True and 0 < row["strict_p50_ms"] < 100 and 0 < row["fast_p50_ms"] < 100
            if depth not in ("7", "8"):
                assert row["same_first_control"] is True and row["cpu_ms"] > row["strict_p50_ms"]
        assert sc["depth"]["8"]["leaves"] == 5172889071            # beyond one expansion: mpc_hip_solve_tree_deep, in pieces
        # the whole depth-8 search fits the reference's 0.1-s budget on one GPU, in both expansions
        assert sc["deepest_level_inside_budget"]["fast"]["levels"] >= 8 and sc["deepest_level_inside_budget"]["strict"]["levels"] >= 8
        assert sc["cpu_deepest_level_inside_budget"] <= 5
    assert t["scenarios"]["lrl_semicircle"]["depth"]["4"]["first_control"] == [-0.0625, 1.0]     # SURVEY.md 8c's probe of the reference


def test_a_region_that_is_no_whole_number_of_launches_takes_a_smaller_group():
    d = _bench("--steps", "41", "--warmup", "5", "--no-cpu-baseline", "--no-tree")
    assert d["config"]["pipeline_level"] == 1 and d["config"]["solves_per_launch"] == 1
    assert d["roofline"]["kernel"] == "rollout_kernel" and d["roofline"]["algorithmic_flop_per_launch"] == 65536 * 20 * (54 * 20 + 21)
    d = _bench("--steps", "42", "--warmup", "5", "--no-cpu-baseline", "--no-tree")       # 42 = 14 x 3
    assert d["config"]["pipeline_level"] == 3 and d["config"]["solves_per_launch"] == 3 and d["roofline"]["kernel"] == "rollout_group_kernel"


def test_two_kernel_form_and_rccl_rehearsal_choose_the_same_control():
    a = _bench("--steps", "40", "--warmup", "5", "--no-cpu-baseline", "--no-tree", "--no-pipeline")
    b = _bench("--steps", "40", "--warmup", "5", "--no-cpu-baseline", "--no-tree", "--force-collective", "--combine-batch", "16")
    assert a["chosen"]["ordinal"] == b["chosen"]["ordinal"] == b["chosen"]["global_ordinal_after_combine"]
    assert (a["chosen"]["rudder"], a["chosen"]["throttle"]) == (b["chosen"]["rudder"], b["chosen"]["throttle"])
    assert b["rccl_ranks"] == 1 and a["rccl_ranks"] is None and b["chosen"]["cross_shard_near_tie_rows"] == 0
    assert a["config"]["deferred_finish"] is False


@pytest.mark.parametrize("workload", ["config4", "config5"])
def test_other_workloads_run(workload):
    d = _bench("--workload", workload, "--steps", "3", "--warmup", "1", "--no-cpu-baseline")
    assert d["value"] > 1e11 and workload[-1] in d["config"]["workload"].split(":")[0]


def test_two_ranks_on_one_gpu_over_gloo_rehearsal():
    """`bench.py --gpus 2 --backend gloo --same-device`: the N-rank bench path (self-launch, barriers, max over ranks,
    ShardCombiner's batched exchange and exact combine) end to end on a 1-GPU box.  The ranks time-share cuda:0, so the
    line is a rehearsal, and says so; what is asserted is the plumbing and the choice."""
    a = _bench("--steps", "20", "--warmup", "4", "--no-cpu-baseline", "--no-tree", "--repeats", "5")
    d = _bench("--gpus", "2", "--backend", "gloo", "--same-device", "--steps", "20", "--warmup", "4", "--repeats", "5",
               "--combine-batch", "8")
    assert d["n_gpus"] == 2 and d["ranks"] == 2 and d["backend"] == "gloo" and d["rccl_ranks"] is None
    assert d["rehearsal"] and "NOT a measurement" in d["rehearsal"]
    assert d["scaling"] == "weak" and d["config"]["candidates_per_gpu"] == 65536
    assert abs(d["value"] - 2 * 65536 * 400 * 20 / (d["timed_region_ms"] * 1e-3)) <= 1e-6 * d["value"]
    # rank 0's shard is the one-GPU run's set; the combined winner is the better of the two shards'
    assert d["chosen"]["global_ordinal_after_combine"] is not None
    assert d["chosen"]["ordinal"] == a["chosen"]["ordinal"]
    assert d["chosen"]["cross_shard_near_tie_rows"] == 0
