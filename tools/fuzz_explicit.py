"""A long random run of EXPLICIT candidate sets against the oracle (python tools/fuzz_explicit.py [problems] [seed]): the
random problems of tests/test_gpu_random_problems.py with arbitrary FP32 controls -- not on the granularity grid, a tenth
of them beyond [-1, 1] / [0, 1] (VehicleState::simulate clamps them, VehicleState.h:44-47) -- ragged set sizes, both
precisions, both selections."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
import mpc_amd as M
from helpers import oracle_explicit, rel_err
import test_gpu_random_problems as T


# the set sizes of the long run; the slice in tests/test_gpu_fuzz_slice.py passes smaller ones
SIZES = (1, 63, 64, 65, 300, 1000, 4097)


def run(n_problems=300, seed=3, solver=None, verbose=True, sizes=SIZES):
    """-> dict(problems, wrong = wrong choices that do NOT carry libm_tie, bad = [findings], libm_ties, worst = {precision: error
    relative to max(|cost|, 1 % of the weights' sum)})"""
    rng = np.random.RandomState(seed)
    solver = solver or M.Solver(0)
    bad, wrong, flagged = [], 0, []
    worst = {M.PREC_F64: 0.0, M.PREC_FAST: 0.0}
    t0 = time.time()
    for k in range(n_problems):
        p = T._problem(rng)
        n = int(rng.choice(sizes))
        rud = rng.uniform(-1.0, 1.0, size=(p.depth, n)).astype(np.float32)
        thr = rng.uniform(0.0, 1.0, size=(p.depth, n)).astype(np.float32)
        wild = rng.uniform(size=(p.depth, n)) < 0.1
        rud[wild] = rng.uniform(-3.0, 3.0, size=int(wild.sum())).astype(np.float32)
        thr[wild] = rng.uniform(-1.0, 2.0, size=int(wild.sum())).astype(np.float32)
        cumulative = bool(rng.randint(0, 2))
        o_last, o_cum, o_best = oracle_explicit(p, rud.astype(np.float64), thr.astype(np.float64), cumulative)
        d_r = torch.tensor(rud, device="cuda").contiguous(); d_t = torch.tensor(thr, device="cuda").contiguous()
        floor = 0.01 * (p.w_distance + p.w_heading + p.w_speed)
        for prec in (M.PREC_F64, M.PREC_FAST):
            if prec == M.PREC_FAST and max(p.substeps[i] for i in range(p.n_substeps)) > 0.1000001:
                continue
            last = torch.empty(n, dtype=torch.float64, device="cuda"); cum = torch.empty(n, dtype=torch.float64, device="cuda")
            res = solver.eval_candidates(p, d_r, d_t, n, prec, M.SELECT_CUMULATIVE if cumulative else M.SELECT_LAST_STEP, last, cum)
            g = (cum if cumulative else last).cpu().numpy()
            o_sel = o_cum if cumulative else o_last
            gl = last.cpu().numpy()
            ok = np.isfinite(o_last) & np.isfinite(gl)
            e = float((np.abs(gl[ok] - o_last[ok]) / np.maximum(np.abs(o_last[ok]), floor)).max()) if ok.any() else 0.0
            worst[prec] = max(worst[prec], e)
            if res.best_ordinal != o_best.best_index:
                rec = (k, prec, n, p.depth, p.n_substeps, cumulative, int(res.best_ordinal), int(o_best.best_index),
                       float(g[res.best_ordinal]), float(g[o_best.best_index]), float(o_sel[res.best_ordinal]), float(o_sel[o_best.best_index]))
                if res.libm_tie:
                    flagged.append(rec)                # the result says the choice is libm-dependent (mpc_hip.h)
                else:
                    wrong += 1
                    bad.append(rec)
            elif e > T.TOL[prec] or not np.array_equal(np.isfinite(gl), np.isfinite(o_last)):
                bad.append((k, prec, n, p.depth, p.n_substeps, "error", e))
        if verbose and k % 50 == 49:
            print("%d problems, wrong choices %d, other findings %d, worst strict %.3g fast %.3g, %.0f s" % (
                k + 1, wrong, len(bad) - wrong, worst[M.PREC_F64], worst[M.PREC_FAST], time.time() - t0), flush=True)
    return {"problems": n_problems, "wrong": wrong, "bad": bad, "libm_ties": flagged, "worst": worst}


if __name__ == "__main__":
    out = run(int(sys.argv[1]) if len(sys.argv) > 1 else 300, int(sys.argv[2]) if len(sys.argv) > 2 else 3)
    print("explicit sets:", out["problems"], "problems; wrong choices", out["wrong"], "; choices that differ and carry libm_tie", len(out["libm_ties"]),
          "; worst error relative to max(|cost|, 1 %% of the weights' sum): strict %.3g fast %.3g" % (out["worst"][M.PREC_F64], out["worst"][M.PREC_FAST]))
    for b in out["bad"] + out["libm_ties"]:
        print("   ", b)
