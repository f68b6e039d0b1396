"""Determinism soak on the GPU: 20,000 config-2 solves, 150 tree searches per expansion and 50 fleet batches must each give
one result, bit for bit (races in the staging / argmin / finish paths would show up as stray winners).  python tools/soak.py"""
import sys
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, mpc_amd as M
from mpc_amd import workloads as W
depth, count = 20, 65536
stream = torch.cuda.Stream()
with torch.cuda.stream(stream):
    s = M.Solver(0, stream=stream.cuda_stream); s.set_timing(0)
    prob = W.config2_problem(depth)
    r, t = W.seeded_controls(W.SEED, 0, count, depth, prob.last_rudder, prob.last_throttle)
    d_r = torch.tensor(r, dtype=torch.float32, device="cuda"); d_t = torch.tensor(t, dtype=torch.float32, device="cuda")
    d_last = torch.empty(count, dtype=torch.float64, device="cuda"); d_cum = torch.empty(count, dtype=torch.float64, device="cuda")
    s.upload([prob], [W.SEED])
    N = 20000
    # the deferred finish (argv[1] == "pipeline"): every solve's finish rides in the next solve's rollout launch, reading
    # one half of the double-buffered working memory while the rollouts fill the other
    # "pipeline2": also two solves per launch (level 2)
    LEVEL = {"pipeline": 1, "pipeline2": 2}.get(sys.argv[1] if len(sys.argv) > 1 else "", 0)
    s.set_pipeline(LEVEL)
    sink = torch.zeros(2 * N, dtype=torch.int64, device="cuda")
    ref_last = None
    for i in range(N):
        s.set_winner_sink(sink[2 * i:2 * i + 2])
        s.run_candidates(d_r, d_t, count, M.PREC_FAST, M.SELECT_LAST_STEP, d_last, d_cum)
        if i % 5000 == 0:
            s.flush()                                # (level 2: a run without a partner has not been launched yet)
            torch.cuda.synchronize()
            cur = d_last.clone()
            if ref_last is None: ref_last = cur
            assert torch.equal(cur, ref_last), i
    s.set_winner_sink(None)
    s.synchronize()                                  # launches the last solve's pending finish
    s.set_pipeline(False)
    torch.cuda.synchronize()
    w = sink.view(N, 2).cpu().numpy()
    print("config2 x", N, "distinct winners:", len(np.unique(w, axis=0)), w[0], flush=True)
    assert len(np.unique(w, axis=0)) == 1
    # pipelined, alternating candidate sources and sizes: every kind of hand-over between consecutive launches
    s.set_pipeline(max(LEVEL, 1))
    M2 = 6000
    sink2 = torch.zeros(2 * M2, dtype=torch.int64, device="cuda")
    kinds = 4
    for i in range(M2):
        s.set_winner_sink(sink2[2 * i:2 * i + 2])
        k = i % kinds
        if k == 0: s.run_candidates(d_r, d_t, count, M.PREC_FAST)
        elif k == 1: s.run_seeded(0, count, M.PREC_FAST)
        elif k == 2: s.run_seeded(1000, 4096, M.PREC_FAST)
        else: s.run_candidates(d_r, d_t, 30000, M.PREC_FAST)
    s.set_winner_sink(None)
    s.synchronize()
    s.set_pipeline(False)
    w2 = sink2.view(M2 // kinds, kinds, 2).cpu().numpy()
    for k in range(kinds):
        u = np.unique(w2[:, k], axis=0)
        print("pipelined mix, kind", k, "distinct winners:", len(u), u[0], flush=True)
        assert len(u) == 1
    # tree, both expansions
    q, _ = W.config1_problem(depth=6)
    for prec in (M.PREC_FAST, M.PREC_F64):
        seen = set()
        for i in range(150):
            res, lv = s.solve_tree(q, 6, precision=prec)
            seen.add((res.best_ordinal, res.best_last, res.best_cumulative, tuple(l.best_leaf for l in lv[:6])))
        print("tree prec", prec, "x150 distinct results:", len(seen), flush=True)
        assert len(seen) == 1
    # fleet batch
    probs, seeds = W.fleet_problems(256, 20)
    seen = set()
    for i in range(50):
        res = s.eval_seeded_batch(probs, seeds, 8192, M.PREC_FAST)
        seen.add(tuple((r.best_ordinal, r.best_last) for r in res))
    print("fleet 256 x 8192 x50 distinct results:", len(seen), flush=True)
    assert len(seen) == 1
print("soak ok")
