"""World size 2 with REAL kernels on both ranks, on one GPU (controller.cpp:158-166 across shards).

Two fresh child processes (tests/two_rank_worker.py), both on cuda:0, process group gloo: RCCL refuses two ranks on one
device, gloo moves host memory, so ShardCombiner stages its one exchange per batch through pinned host buffers.  What
is asserted is what an 8-GPU node must deliver: the winner combined over the ranks == the winner of ONE launch over the
whole set == the oracle's first strict minimum, identically on every rank, at every pipeline level; and for a constructed
pair whose fast scores order the ranks the wrong way round, the exact combine decides as the reference's arithmetic does.
"""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WORKER = os.path.join(ROOT, "tests", "two_rank_worker.py")


def _run(case, tmp_path, extra=()):
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    procs, outs = [], []
    for rank in range(2):
        out = tmp_path / ("%s_%d.json" % (case, rank))
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE="2", LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, "-u", WORKER, case, str(out)] + list(extra), env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
        outs.append(out)
    logs = []
    try:
        for p in procs:
            logs.append(p.communicate(timeout=240)[0])
    finally:
        for p in procs:                              # the exact children started above, nothing else
            if p.poll() is None:
                p.kill()
    for p, log in zip(procs, logs):
        assert p.returncode == 0, log[-3000:]
    return [json.load(open(o)) for o in outs]


def test_two_ranks_combine_to_the_whole_sets_winner(tmp_path):
    r0, r1 = _run("shards", tmp_path)
    assert (r0["world"], r1["world"]) == (2, 2)
    one = r0["one_launch"]
    assert r1["one_launch"]["bits"] == one["bits"] and r1["one_launch"]["ordinal"] == one["ordinal"]
    # the oracle's first strict minimum over all 131,072 candidates
    assert one["ordinal"] == r0["oracle"]["ordinal"]
    assert abs(one["last"] - r0["oracle"]["last"]) <= 1e-5 * r0["oracle"]["last"]
    for level in ("0", "1", "2"):
        for r in (r0, r1):                           # every rank decides alike, at every pipeline level
            c = r["combined"][level]
            assert (c["bits"], c["ordinal"]) == (one["bits"], one["ordinal"]), (level, r["rank"], c, one)
            assert c["partial"] == [one["bits"], one["ordinal"]] and c["rows_alike"] and c["empty_tail"]
    # the combined winner is one of the two shards' winners: the smaller (score bits, global ordinal)
    assert min((r0["shard"]["bits"], r0["shard"]["ordinal"]), (r1["shard"]["bits"], r1["shard"]["ordinal"])) == (one["bits"], one["ordinal"])
    assert r0["shard"]["ordinal"] < 65536 <= r1["shard"]["ordinal"]


def test_cross_rank_near_tie_is_decided_by_the_strict_model(tmp_path):
    r0, r1 = _run("flip", tmp_path)
    assert (r0["a"], r0["b"]) == (r1["a"], r1["b"])
    for i in range(2):
        g0, g1 = r0["rounds"][i], r1["rounds"][i]
        assert g0["fast"] == g1["fast"] and g0["exact"] == g1["exact"]          # identical on both ranks
        rank_a = g0["rank_of_a"]
        assert g0["fast"]["rank"] == 1 - rank_a                                 # the fast keys prefer b ...
        assert g0["fast"]["bits"] == r0["fast_b_bits"]
        assert g0["exact"]["rank"] == rank_a                                    # ... the reference's arithmetic a
        assert g0["exact"]["bits"] == r0["strict_a_bits"]
        assert g0["tied_rows"] == g1["tied_rows"] == i + 1                      # one near-tie row per round, on each rank's handle


def test_two_ranks_search_depth_8_together(tmp_path):
    """the tree search over ranks (distributed.solve_tree_sharded), one shard per rank and two pieces per rank: the combined
    depth-8 record (5.2e9 leaves) is the one mpc_hip_solve_tree_deep finds on a single handle, on both ranks, and exactly one
    rank holds the winner's result"""
    r0, r1 = _run("tree", tmp_path)
    assert r0["deep"] == r1["deep"] and r0["deep"]["levels"] == 8 and r0["deep"]["leaves"] == 5172889071
    for key in ("ppr1", "ppr2"):
        for r in (r0, r1):
            got = {k: r[key][k] for k in ("levels", "leaves", "ordinal", "bits")}
            assert got == r0["deep"], (key, r["rank"], got, r0["deep"])
        assert r0[key]["owner"] == r1[key]["owner"]
        assert [r0[key]["mine_is_winner"], r1[key]["mine_is_winner"]].count(True) == 1
