#!/usr/bin/env python3
"""Digests of what the library returns, bit for bit, on a fixed set of workloads -- the fixture behind "the default
hull reproduces the results the library gave before mpc_hip_set_model existed" (tests/test_gpu_model.py).

    python tools/gpu_bits.py [--lib path/to/libmpc_hip.so] > tests/golden/gpu_bits_default_hull.json

tests/golden/gpu_bits_default_hull.json was written by this script on an MI355X with --lib pointing at a build of
commit 74424a6 (the last one with compile-time model constants); its strict and tree entries still are those digests.
The fast candidate-set entries were re-recorded with the current library when the fast model's spelling changed (the
file's "provenance" says which and why).  The test runs the same function on the current library.  Every value is a SHA-256 of raw float64/uint64 bytes, or an exact integer."""
import argparse
import hashlib
import json
import math
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _sha(*arrays):
    h = hashlib.sha256()
    for a in arrays:
        h.update(np.ascontiguousarray(a).tobytes())
    return h.hexdigest()


def _traj(res):
    names = ("x", "y", "heading", "speed", "rpm", "course_made_good", "speed_over_ground", "rudder", "throttle", "score")
    return np.array([[getattr(res.trajectory[i], k) for k in names] for i in range(res.levels + 1)], dtype=np.float64)


def digests(M, W, setup=None):
    """setup(solver): called once on the fresh handle (the test passes mpc_hip_set_model with the default hull)"""
    import torch
    solver = M.Solver(0)
    if setup:
        setup(solver)
    out = {}

    def seeded(name, p, n, prec, sel=0):
        last = torch.empty(n, dtype=torch.float64, device="cuda")
        cum = torch.empty(n, dtype=torch.float64, device="cuda")
        r = solver.eval_seeded(p, W.SEED, 0, n, prec, sel, last, cum)
        out[name] = {"costs": _sha(last.cpu().numpy(), cum.cpu().numpy()), "ordinal": int(r.best_ordinal),
                     "result": _sha(np.array([r.best_last, r.best_cumulative, r.first_rudder, r.first_throttle]), _traj(r))}

    p2 = W.config2_problem()
    seeded("config2_fast_65536", p2, 65536, M.PREC_FAST)
    seeded("config2_strict_65536", p2, 65536, M.PREC_F64)
    seeded("config2_fast_cumulative_8192", p2, 8192, M.PREC_FAST, 1)
    seeded("config3_strict_8192", W.config3_problem(), 8192, M.PREC_F64)
    seeded("config3_fast_8192", W.config3_problem(), 8192, M.PREC_FAST)
    # the reference's own 11-substep level (ten of 0.1 s and one of 1.4e-16 s) and a ragged schedule
    p11 = M.make_problem((0.5, -0.5, 1.5, 1.8), W.config2_plan().checkpoints(W.T0, 12), M.substep_schedule(1.0),
                         last=(0.1, 0.6), current=(0.2, -0.1))
    seeded("native_schedule_fast_4096", p11, 4096, M.PREC_FAST)
    seeded("native_schedule_strict_4096", p11, 4096, M.PREC_F64)
    prag = M.make_problem((0.5, -0.5, 1.5, 1.8), W.config2_plan().checkpoints(W.T0, 6), [0.1, 0.05, 0.02, 0.1, 0.03],
                          last=(0.1, 0.6), current=(0.2, -0.1))
    seeded("ragged_schedule_fast_4096", prag, 4096, M.PREC_FAST)
    seeded("ragged_schedule_strict_4096", prag, 4096, M.PREC_F64)
    # explicit FP32 controls, deeper than one staging chunk
    p33 = W.config2_problem(depth=31)
    rud, thr = W.seeded_controls(W.SEED, 0, 3000, 31, p33.last_rudder, p33.last_throttle)
    d_r = torch.tensor(rud.astype(np.float32), device="cuda").contiguous()
    d_t = torch.tensor(thr.astype(np.float32), device="cuda").contiguous()
    for name, prec in (("explicit_fast_3000", M.PREC_FAST), ("explicit_strict_3000", M.PREC_F64)):
        last = torch.empty(3000, dtype=torch.float64, device="cuda")
        cum = torch.empty(3000, dtype=torch.float64, device="cuda")
        r = solver.eval_candidates(p33, d_r, d_t, 3000, prec, 0, last, cum)
        out[name] = {"costs": _sha(last.cpu().numpy(), cum.cpu().numpy()), "ordinal": int(r.best_ordinal), "result": _sha(_traj(r))}
    # fleet
    probs, seeds = W.fleet_problems(64, depth=10)
    res = solver.eval_seeded_batch(probs, seeds, 1024, M.PREC_FAST)
    out["fleet_fast_64x1024"] = {"ordinals": [int(r.best_ordinal) for r in res],
                                 "result": _sha(np.array([[r.best_last, r.best_cumulative] for r in res]), *[_traj(r) for r in res])}
    # the reference's tree, both expansions
    pt, _ = W.config1_problem(depth=5)
    w = W.DubinsWrapper((0.0, 0.0, math.pi / 2), (0.0, math.pi, 0.0), 8.0, W.LRL, 2.5, 4.0)
    w.end_time = 4.0 + 6.5
    plan = W.DubinsPlan([w])
    ps = M.make_problem((0.0, 0.0, 0.0, 2.5), plan.checkpoints(4.0, 5), M.substep_schedule(1.0), last=(0.0, 0.0), current=(0.2, -0.1))
    for name, prob in (("tree_straight", pt), ("tree_semicircle", ps)):
        for pname, prec in (("strict", M.PREC_F64), ("fast", M.PREC_FAST)):
            r, lv = solver.solve_tree(prob, 5, precision=prec)
            out["%s_%s" % (name, pname)] = {
                "levels": [[int(lv[l].n_leaves), int(lv[l].best_leaf)] for l in range(r.levels)],
                "result": _sha(np.array([[lv[l].best_score, lv[l].best_cumulative] for l in range(r.levels)]), _traj(r))}
    # one sequence (VehicleState::simulate itself), both models
    for pname, prec in (("strict", M.PREC_F64), ("fast", M.PREC_FAST)):
        r = solver.rollout(p11, [0.3, -0.2, 1.5, 0.0, -1.0, 0.25] * 2, [1.0, 0.5, -0.2, 0.75, 0.0, 1.2] * 2, prec)
        out["rollout_%s" % pname] = _sha(_traj(r))
    # fleet current estimation (strict replay, no current)
    hist = []
    for k in range(5):
        h = []
        for i in range(40 + 7 * k):
            h.append((1.0 * k + 0.2 * i, -2.0 * k + 0.1 * i, (0.3 * k + 0.01 * i) % (2 * math.pi), 1.0 + 0.02 * i, 50.0 + 0.1 * i,
                      0.1 * k - 0.2, 0.5 + 0.1 * k))
        hist.append(h)
    cur = solver.estimate_currents(hist, [(h[-1][0] + 0.5, h[-1][1] - 0.25, h[-1][4] + 0.1) for h in hist])
    out["estimate_currents"] = _sha(np.array(cur))
    solver.close()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--lib", default=None)
    args = ap.parse_args()
    import mpc_amd as M
    from mpc_amd import binding, workloads as W
    if args.lib:
        # an older build lacks the newest entry points: let the binding set signatures on what exists
        import ctypes

        class Lenient(ctypes.CDLL):
            def __getattr__(self, name):
                try:
                    return super().__getattr__(name)
                except AttributeError:
                    if name.startswith("mpc_hip_"):
                        return type("Missing", (), {"argtypes": None, "restype": None})()
                    raise
        real, ctypes.CDLL = ctypes.CDLL, Lenient
        try:
            binding._lib = binding.load_library(os.path.abspath(args.lib))
        finally:
            ctypes.CDLL = real
    json.dump({"library": args.lib or "mpc_amd/libmpc_hip.so", "digests": digests(M, W)}, sys.stdout, indent=1, sort_keys=True)
    sys.stdout.write("\n")


if __name__ == "__main__":
    main()
