"""A fixed-seed slice of the long random runs of tools/fuzz_*.py, so that the driver-run record carries them (the long
runs -- 8,400 trees, 3,000 + 3,000 problems, 1,000 shard splits -- are builder-run: profiles/r0?_fuzz_*.txt):

  * 200 random tree searches of depth 2-4 against the oracle's DFS, both expansions; a chosen leaf may differ from the
    oracle's only where the level carries libm_tie (mpc_hip.h); 150 more with a random boat per problem
  * 200 random problems with arbitrary explicit FP32 controls (a tenth out of range), ragged set sizes, both precisions,
    both selections: no wrong choice, errors within the contract (relative to max(|cost|, 1 % of the weights' sum))
  * 100 random problems cut into 2-8 shards of random sizes: the exact cross-shard decision is the whole set's winner
    and the oracle's

and, by name, the degenerate-weight problems of profiles/r02_fuzz_random_problems.txt with the tolerance they need
written next to them.
"""
import os
import sys

import numpy as np
import pytest

import mpc_amd as M
from helpers import oracle_seeded, rel_err

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


def test_tree_slice(solver):
    import fuzz_tree
    out = fuzz_tree.run(200, seed=20263, solver=solver, verbose=False)
    assert out["bad"] == [], out["bad"][:5]
    assert out["worst"] <= 1e-12
    print("tree slice: 200 problems, flagged libm ties:", len(out["libm_ties"]), "worst level-score error %.3g" % out["worst"])


def test_tree_slice_with_random_hulls(solver):
    """the same with a boat of its own per problem (150 kg - 6 t, 0.5 - 20 kW ...): the final level leaves out nodes whose
    children cannot come near enough to win, by a bound on how far THIS hull can travel in one level (mpc_tree.hip:
    node_margin) -- a bound on the wrong side would lose the oracle's winner here"""
    import fuzz_tree
    out = fuzz_tree.run(150, seed=20266, solver=solver, verbose=False, hulls=True)
    assert out["bad"] == [], out["bad"][:5]
    # (plain relative error of a level's best score: where one weight alone is non-zero the score is a lone difference that
    # cancels to 1e-4 of its operands -- the long runs of profiles/r03_fuzz_tree.txt show up to 1e-9 there, never a wrong leaf)
    assert out["worst"] <= 1e-9
    print("tree slice, random hulls: 150 problems, flagged libm ties:", len(out["libm_ties"]), "worst level-score error %.3g" % out["worst"])


def test_tree_slice_through_the_final_levels_short_list():
    """a wide final level is expanded from a short list of its tiles (mpc_tree.hip: tile_select_kernel); the random trees are
    too shallow to be wide, so a child process runs 120 of them (random boats) with the library's diagnostic switches that
    send every level the synchronised way and every final level through the list"""
    import subprocess
    env = dict(os.environ, MPC_HIP_TREE_SYNC_LEVELS="1", MPC_HIP_TREE_LIST_MIN="1")
    r = subprocess.run([sys.executable, "-u", os.path.join(ROOT, "tools", "fuzz_tree.py"), "120", "20267", "hulls"], env=env,
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=280)
    assert r.returncode == 0, r.stdout[-3000:]
    summary = [l for l in r.stdout.splitlines() if l.startswith("tree problems")]
    assert summary and "tree problems 120 ; disagreements WITHOUT the libm_tie flag 0 ;" in summary[0], r.stdout[-3000:]


def test_tree_slice_through_the_pieces():
    """mpc_hip_solve_tree_deep searches levels beyond one expansion in pieces that resume from the stored levels, share one
    bound and look for the level's one winner (mpc_tree.hip: TreeResume).  Real trees reach that at depth 8; here a child
    process lets the plain search end at level 2 (MPC_HIP_TREE_DEEP_FROM) so that levels 3 and 4 of 100 random trees with
    random boats go through the pieces -- against the oracle's enumeration"""
    import subprocess
    env = dict(os.environ, MPC_HIP_TREE_SYNC_LEVELS="1", MPC_HIP_TREE_LIST_MIN="1", MPC_HIP_TREE_DEEP_FROM="2")
    r = subprocess.run([sys.executable, "-u", os.path.join(ROOT, "tools", "fuzz_tree.py"), "100", "20268", "hulls", "deep"], env=env,
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=280)
    assert r.returncode == 0, r.stdout[-3000:]
    summary = [l for l in r.stdout.splitlines() if l.startswith("tree problems")]
    assert summary and "tree problems 100 ; disagreements WITHOUT the libm_tie flag 0 ;" in summary[0], r.stdout[-3000:]


def test_explicit_slice(solver):
    import fuzz_explicit
    out = fuzz_explicit.run(200, seed=20264, solver=solver, verbose=False, sizes=(1, 63, 64, 65, 300, 1000))
    # the contract of include/mpc_hip.h: relative to max(|cost|, 1 % of the weights' sum).  Strict: 1e-12 holds for the 400
    # Euler steps config 3 is specified on; the generator also draws up to 24 x 64 = 1536 steps, where the long run measured
    # 1.24e-12 -- hence 2e-12 here, as measured, not as contracted.
    assert out["wrong"] == 0, out["bad"][:5]
    assert out["worst"][M.PREC_F64] <= 2e-12 and out["worst"][M.PREC_FAST] <= 1e-5, out["worst"]
    assert [b for b in out["bad"] if b[5] != "error"] == []
    print("explicit slice:", out["worst"], "flagged:", len(out["libm_ties"]))


def test_shard_slice(solver):
    import fuzz_shards
    out = fuzz_shards.run(100, seed=20265, solver=solver, verbose=False, sizes=(512, 1000))
    assert out["bad"] == [] and out["problems"] >= 60, out
    print("shard slice:", out["problems"], "problems, rows with a cross-shard near-tie:", out["tied_rows"])


# (block, k) of tests/test_gpu_random_problems.py's generator -> what the problem is and the PLAIN relative tolerance its
# costs need, per precision (None: the contract's 1e-12 / 1e-5 holds).  All of them have ONE non-zero weight, so a cost is
# a lone |reference - actual| that cancels to ~1e-4 of its operands; against max(|cost|, 1 % of the weights' sum) every one
# of them is within 1.8e-13 (strict) / 2.9e-7 (fast), and none of them chooses wrongly.
DEGENERATE = {
    (145, 5): ("one weight, 22 x 11 substeps", {M.PREC_F64: 2e-12}),
    (175, 9): ("one weight, 16 x 25 substeps", {M.PREC_F64: 4e-12, M.PREC_FAST: 2e-5}),
    (207, 2): ("one weight, 19 x 64 substeps", {M.PREC_F64: 1e-10}),
    (292, 2): ("one weight, 24 x 3 substeps", {M.PREC_F64: 2e-12}),
    (325, 3): ("one weight, 24 x 9 substeps", {M.PREC_F64: 4e-12}),
    (336, 2): ("one weight, 3 x 10 substeps", {M.PREC_F64: 4e-12}),
}


@pytest.mark.parametrize("block,k", sorted(DEGENERATE))
def test_degenerate_weight_regressions(solver, block, k):
    import torch
    import test_gpu_random_problems as T
    rng = np.random.RandomState(1000 + block)
    for j in range(k + 1):                                   # the generator's stream up to problem k of the block
        p = T._problem(rng)
        seed = int(rng.randint(1, 2 ** 31))
        first = int(rng.randint(0, 5000))
    n = 1024
    what, tol = DEGENERATE[(block, k)]
    assert sum(1 for w in (p.w_distance, p.w_heading, p.w_speed) if w > 0) == 1, what
    o_last, o_cum, o_best = oracle_seeded(p, seed, first, n)
    floor = 0.01 * (p.w_distance + p.w_heading + p.w_speed)
    for prec in (M.PREC_F64, M.PREC_FAST):
        if prec == M.PREC_FAST and max(p.substeps[i] for i in range(p.n_substeps)) > 0.1000001:
            continue
        last = torch.empty(n, dtype=torch.float64, device="cuda")
        res = solver.eval_seeded(p, seed, first, n, prec, M.SELECT_LAST_STEP, last, None)
        g = last.cpu().numpy()
        assert res.best_ordinal == o_best.best_index, (what, prec)                      # never a wrong choice
        plain = float(rel_err(g, o_last).max())
        assert plain <= tol.get(prec, T.TOL[prec]), (what, prec, plain)               # the tolerance it actually needs
        floored = float((np.abs(g - o_last) / np.maximum(np.abs(o_last), floor)).max())
        assert floored <= (1e-12 if prec == M.PREC_F64 else 1e-5), (what, prec, floored)   # the contract as mpc_hip.h words it
