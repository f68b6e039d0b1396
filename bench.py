#!/usr/bin/env python3
"""bench.py -- candidate-rollout steps/s of the MPC hot path on MI355X (BASELINE.json metric, config 2).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

A "step" is one MPC solve over one batch of candidates with everything resident in HBM: the rollout kernel
(65,536 candidates x 20 levels x 20 Euler substeps + tracking cost), the argmin, the strict decision among
near-ties and the winner's trajectory.  With N > 1 every rank evaluates its own 65,536-candidate shard of one
N x 65,536 set (weak scaling) and the shards are combined by an RCCL min-allreduce of the
packed (score bits, ordinal) winner.  Rank 0 prints ONE JSON line.

--workload config4 / config5 time the other single-GPU-sized pieces of BASELINE.json (one GPU's 2^21-candidate
shard of the 16 M x 30 s set; 4096/N vessels x 8192 candidates) with the same contract; the default is config 2,
the configuration the metric is quoted on.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

N_CAND = 65536
FLOP_PER_STEP = 54           # VehicleState::estimate as written
FLOP_PER_SCORE = 21          # Controller::compareStates
FP64_VECTOR_PEAK_TFLOPS = 78.6   # MI355X FP64 vector peak = half the 157.3 TFLOP/s FP32 vector peak (MI355X_MICROARCH.md)
HBM_PEAK_GBS = 8000.0


def cpu_baseline(problem, seeds, want_ordinals):
    """The reference-equivalent CPU loop (oracle/, a port of controller.cpp:149-166 + VehicleState.h) timed on
    ONE host thread -- the reference solve is single-threaded -- on the same candidate sets: the sets the timed stream
    rotates through, one after the other (each checked against the GPU's choice for THAT set), until ~10 s are covered."""
    import oracle as O
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from helpers import oracle_args
    cfg, veh, cps, sub = oracle_args(problem)
    reps, t_total, same = 0, 0.0, []
    while (t_total < 10.0 and reps < 8) or reps < len(seeds):
        seed, want = seeds[reps % len(seeds)], want_ordinals[reps % len(seeds)]
        t0 = time.perf_counter()
        _, _, best = O.eval_seeded(cfg, veh, cps, problem.depth, sub, seed, 0, 1, N_CAND)
        t_total += time.perf_counter() - t0
        if reps < len(seeds):
            same.append(bool(best.best_index == want))
        reps += 1
    steps = reps * N_CAND * problem.depth * problem.n_substeps
    # additionally: the same loop on all the host threads this job may use, each on its own slice of
    # the set -- NOT how the reference runs (one std::async thread, controller.cpp:295), labelled as such
    from concurrent.futures import ThreadPoolExecutor
    threads = int(os.environ.get("OMP_NUM_THREADS", "0")) or min(16, os.cpu_count() or 1)
    per = N_CAND // threads
    def slice_(k):                     # ctypes releases the GIL inside the C call
        return O.eval_seeded(cfg, veh, cps, problem.depth, sub, seeds[0], k * per, 1, per)[2]
    t0 = time.perf_counter()
    with ThreadPoolExecutor(threads) as ex:
        list(ex.map(slice_, range(threads)))
    t_all = time.perf_counter() - t0
    return {
        "value": steps / t_total, "unit": "steps/s", "cores": 1, "kind": "port",
        "sample": "%d x a full config-2 set (65536 candidates x 400 steps; the %d sets of the timed stream in turn), %.1f s" % (reps, len(seeds), t_total),
        "same_choice_as_gpu": all(same), "same_choice_per_set": same,
        "host_cpus": os.cpu_count(),
        "all_threads": {"value": threads * per * problem.depth * problem.n_substeps / t_all, "unit": "steps/s",
                        "cores": threads, "note": "not the reference's behaviour (its solve is single-threaded)"},
    }


def cpu_baseline_tree(problem, depth):
    """The oracle's restatement of the reference's own search (Controller::mpc's iterative-deepening DFS,
    controller.cpp:80-189) on one host thread, frozen clock: wall time of the whole deepening loop up to `depth` and the
    control it chooses.  Part of the cpu_baseline leg (the only place bench.py touches the oracle)."""
    import oracle as O
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from helpers import oracle_args
    cfg, veh, cps, sub = oracle_args(problem)
    start = O.State(problem.start_x, problem.start_y, problem.start_heading, problem.start_speed, 0.0)
    t0 = time.perf_counter()
    it, traj, olev, nodes = O.mpc_tree(cfg, start, cps, depth, -1)
    dt = time.perf_counter() - t0
    return {"ms": dt * 1e3, "levels": int(it), "first_control": [traj[1].last_rudder, traj[1].last_throttle],
            "leaves": int(olev[it - 1].n_leaves), "simulate_calls": int(nodes)}


def tree_block(M, W, device, with_cpu):
    """The PRODUCTION path in the bench line: the reference-native limited-branching tree search exactly as the drop-in
    runs it (mpc_hip_solve_tree / _deep through the C-ABI, one synchronous solve at a time, problem upload and result copy
    included), on the reference's own end-to-end scenario (LRL semicircle, test_controller.cpp:449-467) and on BASELINE
    config 1's straight line: p50 wall per depth for both expansions, the deepest level completed inside the reference's
    0.1-s planning budget (controller.h:153), and beside them the oracle's single-thread time and choice."""
    solver = M.Solver(device)
    solver.set_timing(0)
    out = {"note": "p50 wall ms of one synchronous mpc_hip_solve_tree call (host -> C-ABI -> result filled); strict = the reference's "
                   "arithmetic expands the tree, fast = the streamlined model expands and the strict one decides; cpu = the oracle's DFS "
                   "restatement on one host thread (the whole deepening loop to that depth); simulate_calls = nodes of the tree as "
                   "the reference walks them -- on the last level the library simulates only nodes that can still win (same result)",
           "budget_s": 0.1, "scenarios": {}}
    scenarios = {"lrl_semicircle": lambda d: W.semicircle_problem(d + 1.5)[0],
                 "config1_straight_line": lambda d: W.config1_problem(depth=d)[0]}
    for name, make in scenarios.items():
        rows = {}
        for depth in (3, 4, 5, 7, 8):
            p = make(depth)
            assert p.depth == depth, (name, depth, p.depth)
            row = {}
            chosen = {}
            # depth 8 (5.2e9 leaves) is beyond one expansion: mpc_hip_solve_tree_deep, what the drop-in calls, searches it in pieces
            solve = solver.solve_tree_deep if depth > 7 else solver.solve_tree
            for label, prec in (("strict", M.PREC_F64), ("fast", M.PREC_FAST)):
                reps = 40 if depth <= 5 else (7 if depth <= 7 else 3)
                solve(p, depth, precision=prec)                             # warm: buffers grown, code objects loaded
                lat = []
                for _ in range(reps):
                    a = time.perf_counter()
                    res, levels = solve(p, depth, precision=prec)
                    lat.append((time.perf_counter() - a) * 1e3)
                row[label + "_p50_ms"] = float(np.median(lat))
                chosen[label] = (res.first_rudder, res.first_throttle, int(res.best_ordinal), int(res.levels), int(res.libm_tie))
                row["leaves"] = int(levels[depth - 1].n_leaves)
                row["simulate_calls"] = int(res.evaluated)
            row["first_control"] = list(chosen["strict"][:2])
            row["libm_tie"] = bool(chosen["strict"][4])
            row["strict_and_fast_agree"] = chosen["strict"] == chosen["fast"]
            if with_cpu and depth <= 5:                                    # depth 7 is ~150 s on the CPU
                c = cpu_baseline_tree(p, depth)
                row["cpu_ms"] = c["ms"]
                row["cpu_simulate_calls"] = c["simulate_calls"]
                row["same_first_control"] = c["first_control"] == row["first_control"]
            else:
                row["cpu_ms"], row["same_first_control"] = None, None
            rows[str(depth)] = row
        # the deadline as the vessel has it: a real clock, 0.1 s, a plan long enough for depth 9 (8.8e10 leaves: whether the
        # pieces of that level finish inside the budget depends on the scenario and the box; an unfinished level does not count)
        p8 = make(9)
        deepest = {}
        for label, prec in (("strict", M.PREC_F64), ("fast", M.PREC_FAST)):
            got = []
            for _ in range(3):
                t0 = time.perf_counter()
                res, _ = solver.solve_tree_deep(p8, 9, precision=prec, now=time.perf_counter, end_time=t0 + 0.1)
                got.append((int(res.levels), (time.perf_counter() - t0) * 1e3))
            deepest[label] = {"levels": min(g[0] for g in got), "wall_ms": float(np.median([g[1] for g in got]))}
        cpu_deepest = None
        if with_cpu:                                                       # the deepest loop the oracle finishes inside the budget
            cpu_deepest = max([int(d) for d, r in rows.items() if r["cpu_ms"] is not None and r["cpu_ms"] <= 100.0] or [0])
        out["scenarios"][name] = {"depth": rows, "deepest_level_inside_budget": deepest, "cpu_deepest_level_inside_budget": cpu_deepest}
    solver.close()
    return out


def measured_valu(k1_seconds, spl=1):
    """SURVEY.md 8d's second figure: rocprof-counted VALU instructions x 64 lanes / (kernel time x peak lane-issue
    rate 256 CU x 4 SIMD x 16 lanes x 2.4 GHz).  The instruction count comes from the committed PMC pass of this
    command (counters cannot be read inside the timed process), the time is this run's."""
    try:
        prof = json.load(open(os.path.join(ROOT, "profiles", "r03_pmc_summary.json")))
        scale = spl / float(prof.get("solves_per_launch", 1))       # the profile's launch may hold another number of solves
        insts = float(prof["sq_insts_valu"]) * scale
    except Exception:
        return None
    out = {"sq_insts_valu_per_launch": insts, "lane_issue_frac": insts * 64 / (k1_seconds * 256 * 4 * 16 * 2.4e9),
           "source": "profiles/r03_pmc_summary.json (SQ_INSTS_VALU) / this run's kernel time"}
    try:        # hardware-counted FP64 operations (SQ_INSTS_VALU_{FMA,MUL,ADD,TRANS}_F64 x 64 lanes, FMA = 2 FLOP)
        rows = {ln.split("|")[0].strip(): float(ln.split("|")[5])
                for ln in open(os.path.join(ROOT, "profiles", "r03_fp64_flop_counters.txt")) if ln.startswith("config2")}
        # "config2_group": counted on the four-solve launch itself; "config2": on a one-solve launch
        flop = rows["config2_group"] * spl / 4.0 if spl >= 2 and "config2_group" in rows else rows["config2"] * spl
        out["counted_fp64_flop_per_launch"] = flop
        out["counted_fp64_tflops"] = flop / k1_seconds / 1e12
        out["counted_fp64_frac_of_peak"] = flop / k1_seconds / (FP64_VECTOR_PEAK_TFLOPS * 1e12)
    except Exception:
        pass
    return out


def measured_traffic(spl=1):
    """HBM bytes per rollout-kernel launch from the committed PMC profile of this same command (rocprofv3 --pmc
    cannot run inside the timed process); None if no profile has been committed."""
    path = os.path.join(ROOT, "profiles", "r03_pmc_summary.json")
    try:
        prof = json.load(open(path))
        return int(prof["hbm_bytes_per_launch"] * spl / float(prof.get("solves_per_launch", 1)))
    except Exception:
        return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=200)
    ap.add_argument("--ramp-seconds", type=float, default=0.3,
                    help="untimed solves before the warmup steps: the GPU takes ~20 ms of load to reach its sustained "
                         "clocks (200 steps after a 20-step warmup measure 53 us per solve, after 2000 warmup steps 47 us)")
    ap.add_argument("--precision", choices=["fast", "f64"], default="fast")
    ap.add_argument("--workload", choices=["config2", "config4", "config5"], default="config2")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-tree", action="store_true", help="skip the `tree` block (the reference-native search timed through the C-ABI)")
    ap.add_argument("--sets", type=int, default=4,
                    help="config 2: how many DISTINCT candidate sets (seeds SEED, SEED+1, ...) are resident and solved in turn, each with "
                         "its own control arrays and cost outputs -- a stream of different solves, not one solve repeated")
    ap.add_argument("--timing-period", type=int, default=8,
                    help="HIP events ride on every n-th solve's dispatches inside the timed region (1 = every solve)")
    ap.add_argument("--combine-batch", type=int, default=64,
                    help="solves per all-gather of the shard winners (multi-GPU runs)")
    ap.add_argument("--repeats", type=int, default=0,
                    help="how many times the K-step region is timed (the median region is reported); 0 = until 0.5 s are "
                         "covered, at most 2000 regions")
    ap.add_argument("--no-pipeline", action="store_true",
                    help="two kernels per solve (rollouts, finish) instead of the deferred finish (mpc_hip_set_pipeline)")
    ap.add_argument("--pipeline-level", type=int, choices=[0, 1, 2, 3, 4], default=4,
                    help="mpc_hip_set_pipeline: 1 = every solve's finish rides in the next solve's rollout launch; K >= 2 = also K "
                         "solves per launch (a 65,536-candidate set is one wave per SIMD: the second wave comes from the next solve, "
                         "and a launch's fixed ~10 us are paid once per K solves)")
    ap.add_argument("--force-collective", action="store_true",
                    help="run the RCCL combine step even with one rank (rehearsal of the N > 1 path on a 1-GPU box)")
    ap.add_argument("--backend", choices=["nccl", "gloo"], default="nccl",
                    help="torch.distributed backend of the combine step: nccl = RCCL over xGMI (the measured path); gloo = the "
                         "exchange staged through pinned host memory (rehearsals: RCCL refuses two ranks on one device)")
    ap.add_argument("--same-device", action="store_true",
                    help="every rank uses cuda:0 (with --backend gloo: the N-rank pipeline end to end on a 1-GPU box; the ranks "
                         "time-share the GPU, so the line is a functional rehearsal, not a measurement)")
    args = ap.parse_args()

    # --gpus N without a launcher: start the N ranks ourselves, BEFORE anything in this process touches the GPU (a
    # process that has initialised HIP must never be replaced or forked into ranks), and relay rank 0's JSON line
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        import socket
        import subprocess
        with socket.socket() as so:
            so.bind(("127.0.0.1", 0))
            port = so.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        raise SystemExit(subprocess.run(cmd).returncode)

    # ONE JSON line on stdout: RCCL prints a version banner on file descriptor 1 when the first communicator comes up,
    # so everything but our line is sent to stderr at the descriptor level
    sys.stdout.flush()
    json_out = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)

    import torch
    import torch.distributed as dist
    import mpc_amd as M
    from mpc_amd import workloads as W
    from mpc_amd.distributed import ShardCombiner, combine_winners_exact_device

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d: launch with torch.distributed.run --nproc-per-node %d"
                         % (args.gpus, world, args.gpus))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: there is no CPU fallback for the measured path")
    if args.same_device:
        if world > 1 and args.backend == "nccl":
            raise SystemExit("--same-device with several ranks needs --backend gloo (RCCL refuses two ranks on one device)")
        local = 0
    torch.cuda.set_device(local)
    if world > 1 or args.force_collective:
        if world == 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
            os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group("gloo")
    # how many ranks the backend really connected: an all-reduce of ones over the communicator the combine step uses
    coll_dev = "cuda" if args.backend == "nccl" else "cpu"       # where the tensors of the bench's own small collectives live
    rccl_ranks = ranks = None
    if dist.is_initialized():
        ones = torch.ones(1, dtype=torch.int64, device=coll_dev)
        dist.all_reduce(ones)
        ranks = int(ones.item())
        assert ranks == world, (ranks, world)
        rccl_ranks = ranks if args.backend == "nccl" else None

    prec = M.PREC_FAST if args.precision == "fast" else M.PREC_F64
    stream = torch.cuda.Stream()
    with torch.cuda.stream(stream):
        solver = M.Solver(local, stream=stream.cuda_stream)
        combine = world > 1 or args.force_collective
        if args.workload == "config2":
            depth, count, n_problems = 20, N_CAND, 1
            problem = W.config2_problem(depth)
            first = rank * count
            # this rank's shard of every seeded set, as explicit FP32 SoA controls [depth][n] resident in HBM.  The stream
            # rotates through --sets DISTINCT sets (own control arrays, own cost outputs): at pipeline level 2 the two solves
            # of a launch are different candidate sets, as two consecutive solves of a sampling controller would be
            n_sets = max(1, args.sets)
            set_seeds = [W.SEED + k for k in range(n_sets)]
            sets = []
            for sd in set_seeds:
                rud, thr = W.seeded_controls(sd, first, count, depth, problem.last_rudder, problem.last_throttle)
                sets.append((torch.tensor(rud, dtype=torch.float32, device="cuda").contiguous(),
                             torch.tensor(thr, dtype=torch.float32, device="cuda").contiguous(),
                             torch.empty(count, dtype=torch.float64, device="cuda"), torch.empty(count, dtype=torch.float64, device="cuda")))
            d_rud, d_thr, d_last, d_cum = sets[0]
            solver.upload([problem], [W.SEED])
            turn = [0]

            def launch():
                r_, t_, l_, c_ = sets[turn[0] % n_sets]
                turn[0] += 1
                solver.run_candidates(r_, t_, count, prec, M.SELECT_LAST_STEP, l_, c_)
            offset = first
            what = ("BASELINE config 2: 65,536 seeded candidates per GPU x 20 s horizon (20 levels x 20 Euler substeps "
                    "of 0.05 s), curved Dubins reference (LSR+RSL+LRL, rho 8 m, 2.5 m/s), explicit FP32 SoA controls "
                    "resident in HBM")
        elif args.workload == "config4":
            depth, count, n_problems = 30, 1 << 21, 1
            problem = W.config2_problem(depth)
            first = rank * count
            solver.upload([problem], [W.SEED])
            launch = lambda: solver.run_seeded(first, count, prec)
            offset = 0                                     # the library already reports global ordinals
            what = ("BASELINE config 4: 2^21 device-generated candidates per GPU (one shard of the 16 M set at 8 GPUs) x "
                    "30 s horizon (30 x 20 substeps)")
        else:
            depth, count = 20, 8192
            n_problems = 4096 // world
            probs, seeds = W.fleet_problems(4096, depth)
            probs, seeds = probs[rank * n_problems:(rank + 1) * n_problems], seeds[rank * n_problems:(rank + 1) * n_problems]
            problem = probs[0]
            first = 0
            solver.upload(probs, seeds)
            launch = lambda: solver.run_seeded(0, count, prec)
            offset, combine = 0, False                     # problems are sharded: no collective on the data path
            what = "BASELINE config 5: 4096 vessels (sharded over the GPUs) x 8,192 candidates x 20 s horizon"
        # non-negative doubles order like their bit patterns: (score bits, global ordinal) lexicographic minimum over
        # the shards = the reference's first strict minimum.  One async all-gather per --combine-batch solves (fewer, larger
        # collectives), reduced one batch later so that its latency hides behind rollouts (mpc_amd/distributed.py).
        combiner = ShardCombiner(solver, n_problems, ordinal_stride=(count if args.workload == "config2" else 0),
                                 always=args.force_collective, batch=args.combine_batch, depth=depth) if combine else None

        def step():
            if combiner:
                combiner.begin()
            launch()
            if combiner:
                combiner.submit()

        # a stream of solves: each run's finish rides in the next run's rollout launch (one dispatch per solve); the
        # results are bit for bit those of the two-kernel form (tests/test_gpu_pipeline.py)
        level = 0 if args.no_pipeline else args.pipeline_level
        while level >= 2 and args.steps % level:
            level -= 1         # a region that is not a whole number of launches would end in a smaller launch and blur the per-launch kernel time
        solver.set_pipeline(level)
        # solves per rollout launch: levels >= 2 group the one-wave-per-SIMD solves of config 2 (fast precision) and nothing else
        spl = level if (level >= 2 and args.workload == "config2" and prec == M.PREC_FAST) else 1
        solver.set_timing(0)
        t_ramp = time.perf_counter()
        while time.perf_counter() - t_ramp < args.ramp_seconds and args.workload == "config2":
            for _ in range(50):
                launch()
            solver.synchronize()
        solver.set_timing(max(1, min(args.timing_period, args.steps)))
        for _ in range(args.warmup):
            step()
        solver.flush()                     # an odd warmup leaves no run waiting for a partner from the timed region
        solver.timing(reset=True)

        def timed_region():
            """EXACTLY args.steps steps, bracketed by barrier + synchronize on both sides; max over the ranks"""
            if world > 1:
                dist.barrier()
            torch.cuda.synchronize()
            if args.workload == "config2":
                turn[0] = 0                                    # every region solves the same sequence of sets: 0, 1, ..., K-1 mod --sets
            t0 = time.perf_counter()
            for _ in range(args.steps):
                step()
            comb = combiner.flush() if combiner else None      # every solve of the region is combined by here
            solver.flush()                                     # ... and finished (the last solve's deferred finish)
            if world > 1:
                dist.barrier()
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            if world > 1:
                t = torch.tensor([dt], dtype=torch.float64, device=coll_dev)
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                dt = float(t.item())
            return dt, comb

        # A region of a few dozen 37-us solves is under a millisecond -- one scheduler hiccup of the host moves it by
        # several per cent.  So the K-step region is measured `repeats` times (each one bracketed as above) and the MEDIAN
        # region is the one reported: still exactly K steps, no longer a single sub-millisecond sample.
        repeats = args.repeats
        regions = []
        combined = None
        while True:
            dt, combined = timed_region()
            regions.append(dt)
            if (len(regions) >= repeats) if repeats > 0 else (sum(regions) >= 0.5 or len(regions) >= 2000):
                break                                               # auto: at least 0.5 s in all, at most 2000 regions
        # (the ranks agree on the number of regions: every dt above is already the max over the ranks)
        elapsed = float(np.median(regions))
        runs, k1_ms_sum, total_ms_sum = solver.timing(reset=True)
        # for the record: the same regions with ONE solve per launch (pipeline level 1), not part of `value`
        one_per_launch = None
        if spl >= 2:
            solver.set_pipeline(1)
            solver.set_timing(0)
            for _ in range(4):
                step()
            solver.flush()
            alt = [timed_region()[0] for _ in range(9)]
            one_per_launch = {"value": world * n_problems * count * depth * problem.n_substeps * args.steps / float(np.median(alt)),
                              "ms_per_step": float(np.median(alt)) / args.steps * 1e3, "regions": len(alt),
                              "note": "mpc_hip_set_pipeline level 1: deferred finish only"}
            solver.set_pipeline(level)
        res = solver.fetch()[0]
        solver.set_pipeline(False)  # the latency loop is one synchronous solve at a time
        # what the GPU chooses for EVERY set of the stream (one synchronous solve each): the CPU baseline checks each
        set_choices = None
        if args.workload == "config2":
            set_choices = []
            for r_, t_, l_, c_ in sets:
                solver.run_candidates(r_, t_, count, prec, M.SELECT_LAST_STEP, l_, c_)
                q = solver.fetch()[0]
                set_choices.append({"ordinal": int(q.best_ordinal) + offset, "rudder": q.first_rudder, "throttle": q.first_throttle,
                                    "last_score": q.best_last, "libm_tie": bool(q.libm_tie)})
            # `chosen` below: the set the timed region's LAST solve worked on (the one a combine's last row refers to)
            last_set = (args.steps - 1) % n_sets
            r_, t_, l_, c_ = sets[last_set]
            solver.run_candidates(r_, t_, count, prec, M.SELECT_LAST_STEP, l_, c_)
            res = solver.fetch()[0]
        solver.set_timing(0)       # the latency loop below is timed by the host clock: no events on its dispatches (~5 us each)

        # p50 solve latency (BASELINE's second metric): host call -> result struct filled, including the problem upload,
        # the result copy and, with several GPUs, the synchronous combine of the shard winners (two min-allreduces)
        p50 = None
        if args.workload == "config2":
            lat = []
            for _ in range(min(max(args.steps, 20), 200)):
                torch.cuda.synchronize()
                if world > 1:
                    dist.barrier()
                a = time.perf_counter()
                solver.eval_candidates(problem, d_rud, d_thr, count, prec, M.SELECT_LAST_STEP, d_last, d_cum)
                if combine:             # the synchronous form of the exact cross-shard decision (one all-gather of the winner records)
                    combine_winners_exact_device(solver, depth, ordinal_stride=count)[0, 1].item()
                lat.append((time.perf_counter() - a) * 1e3)
            p50 = float(np.median(lat))

    nsub = problem.n_substeps
    steps_per_solve = n_problems * count * depth * nsub
    value = world * steps_per_solve * args.steps / elapsed
    k1_ms = k1_ms_sum / max(runs, 1)
    # per LAUNCH of the dominant kernel: spl solves
    flop = spl * n_problems * count * depth * (FLOP_PER_STEP * nsub + FLOP_PER_SCORE)
    hbm_bytes = spl * n_problems * count * ((2 * 4 * depth if args.workload == "config2" else 0) + 3 * 8)
    if rank == 0:
        line = {
            "metric": "candidate_rollout_steps_per_sec",
            "value": value, "unit": "steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "timed_region_ms": elapsed * 1e3,
            "timed_regions": {"count": len(regions), "reported": "median", "min_ms": min(regions) * 1e3, "max_ms": max(regions) * 1e3,
                              "note": "each region is exactly `steps` steps between barrier + synchronize"},
            "rccl_ranks": rccl_ranks, "ranks": ranks, "backend": (args.backend if dist.is_initialized() else None),
            "rehearsal": ("%d ranks time-share cuda:0 and exchange through host memory (gloo): the N-rank pipeline end to end on "
                          "one GPU -- a functional rehearsal, NOT a measurement" % world) if (args.same_device and world > 1) else None,
            "higher_is_better": True,
            "scaling": "strong" if args.workload == "config5" else "weak",     # config 5 splits a fixed 4096-vessel fleet
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {"workload": what, "candidates_per_gpu": n_problems * count, "problems_per_gpu": n_problems,
                       "depth": depth, "substeps_per_level": nsub,
                       "precision": args.precision + (" (FP64 state, streamlined model, FP32 atan2 in the cost)" if prec == M.PREC_FAST else " (reference statement order)"),
                       "selection": "last_step", "deferred_finish": level >= 1, "pipeline_level": level, "solves_per_launch": spl,
                       "distinct_candidate_sets": (n_sets if args.workload == "config2" else None),
                       "stream": (("%d independent solves share each rollout launch (mpc_hip_set_pipeline level %d): a 65,536-candidate set is one "
                                   "wave per SIMD, the second wave comes from the NEXT solves of the stream -- different candidate sets with their own "
                                   "control arrays and outputs -- and a launch's fixed cost is paid once; `one_solve_per_launch` gives the rate "
                                   "without it" % (spl, spl)) if spl > 1 else
                                  "one solve per rollout launch"),
                       "parallelism": ("candidate shards; per %d solves ONE async RCCL all-gather of the winner records (score, ordinal, controls), then "
                                       "mpc_hip_combine_shards_exact: strict re-score of cross-shard near-ties from the gathered controls, "
                                       "the reference's first strict minimum across GPUs" % args.combine_batch if combine else
                                       ("problem shards, no collective" if world > 1 else "single GPU"))},
            "clock_ramp_s": args.ramp_seconds if args.workload == "config2" else 0.0,
            "p50_solve_latency_ms": p50,
            "one_solve_per_launch": one_per_launch,
            "rollout_kernel_ms": k1_ms,
            "kernel_timing": "HIP events attached to %d rollout dispatches of the %d timed solves (every %d-th solve's; a dispatch "
                             "holds %d solve%s)" % (runs, args.steps * len(regions), max(1, min(args.timing_period, args.steps)), spl,
                                                    "s" if spl > 1 else ""),
            "solve_gpu_ms": total_ms_sum / max(runs, 1),
            "chosen": {"ordinal": int(res.best_ordinal) + offset, "rudder": res.first_rudder,
                       "throttle": res.first_throttle, "last_score": res.best_last,
                       "per_set": set_choices,
                       "global_ordinal_after_combine": (int(combined[1][0].item()) if combined else None),
                       "cross_shard_near_tie_rows": (combiner.tied_rows() if combiner else None)},
            "roofline": {
                "bound": "valu",
                "bound_note": "FP64 vector ALU (SURVEY.md 8d): neither mfma nor hbm -- sequential per-candidate ODE "
                              "integration, no dense contraction; HBM figures under 'hbm'",
                "kernel": "rollout_group_kernel" if spl >= 2 else "rollout_kernel", "solves_per_launch": spl, "achieved": flop / (k1_ms * 1e-3) / 1e12, "peak": FP64_VECTOR_PEAK_TFLOPS,
                "unit": "TFLOP/s", "frac": flop / (k1_ms * 1e-3) / 1e12 / FP64_VECTOR_PEAK_TFLOPS,
                "traffic": measured_traffic(spl) if args.workload == "config2" and prec == M.PREC_FAST else None,
                "traffic_source": "profiles/r03_pmc_summary.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command)",
                "measured_valu": measured_valu(k1_ms * 1e-3, spl) if args.workload == "config2" and prec == M.PREC_FAST else None,
                "algorithmic_flop_per_launch": flop, "kernel_steps_per_sec": spl * steps_per_solve / (k1_ms * 1e-3),
                "hbm": {"algorithmic_bytes_per_launch": hbm_bytes,
                        "achieved_GBps": hbm_bytes / (k1_ms * 1e-3) / 1e9, "peak_GBps": HBM_PEAK_GBS},
            },
        }
        if world == 1 and not args.no_cpu_baseline and args.workload == "config2":
            line["cpu_baseline"] = cpu_baseline(problem, set_seeds, [c["ordinal"] for c in set_choices])
        if world == 1 and not args.no_tree and args.workload == "config2":
            line["tree"] = tree_block(M, W, local, with_cpu=not args.no_cpu_baseline)
        json_out.write(json.dumps(line) + "\n")
        json_out.flush()
    solver.close()
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
