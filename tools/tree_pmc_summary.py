"""Summaries of tools/collect_profiles_tree.sh's passes for the committed evidence:
    python tools/tree_pmc_summary.py gpurun_out/t3 tree   > profiles/r03_tree_pmc_summary.txt
    python tools/tree_pmc_summary.py gpurun_out/t3 strict > profiles/r03_strict_pmc_summary.txt
Per kernel: the counters (mean per launch), the kernel-trace duration, and what follows from them -- VALU instructions
per child (or per substep), the FP64 share, how busy the VALUs were, algorithmic and hardware-counted FLOP/s."""
import csv
import glob
import sys
from collections import defaultdict

src, which = sys.argv[1], sys.argv[2]
PEAK = 78.6e12
SIMDS = 1024


def counters(d, kernel, largest_grid_only=False):
    """mean per launch; largest_grid_only: over the launches with the kernel's largest grid only (the tree: one level)"""
    rows = []
    for path in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        rows += [r for r in csv.DictReader(open(path)) if kernel in r["Kernel_Name"]]
    if largest_grid_only and rows:
        g = max(int(r["Grid_Size"]) for r in rows)
        rows = [r for r in rows if int(r["Grid_Size"]) == g]
    acc = defaultdict(lambda: [0.0, 0])
    for r in rows:
        acc[r["Counter_Name"]][0] += float(r["Counter_Value"]); acc[r["Counter_Name"]][1] += 1
    return {k: s / n for k, (s, n) in acc.items()}


def stats(d, kernel, pick=max):
    """kernel-trace: the LARGEST launch of the kernel for the tree (the depth-7 level), the average for the rollout"""
    for path in glob.glob(d + "/stats/*kernel_stats.csv"):
        for r in csv.DictReader(open(path)):
            if kernel in r["Name"]:
                return float(r["MaxNs"]) if pick is max else float(r["AverageNs"]), int(r["Calls"])
    return None, 0


def report(title, d, kernel, units, unit_name, flop_per_unit, pmc_launch_is_max):
    c = counters(d, kernel, pmc_launch_is_max)
    ns, calls = stats(d, kernel, max if pmc_launch_is_max else None)
    print("## " + title)
    print("# kernel-trace: %s = %.1f us (%d launches in the trace)" % ("largest launch" if pmc_launch_is_max else "average launch", ns / 1e3, calls))
    for k in sorted(c):
        print("%-28s %.6g" % (k, c[k]))
    valu = c["SQ_INSTS_VALU"]
    f64 = c["SQ_INSTS_VALU_FMA_F64"] + c["SQ_INSTS_VALU_MUL_F64"] + c["SQ_INSTS_VALU_ADD_F64"] + c["SQ_INSTS_VALU_TRANS_F64"]
    hw_flop = (2 * c["SQ_INSTS_VALU_FMA_F64"] + c["SQ_INSTS_VALU_MUL_F64"] + c["SQ_INSTS_VALU_ADD_F64"] + c["SQ_INSTS_VALU_TRANS_F64"]) * 64
    alg = units * flop_per_unit
    cycles = c["GRBM_GUI_ACTIVE"] / 8.0            # rocprofv3 sums the counter over the 8 XCDs
    print("# derived:")
    print("#   %s: %.4g;  VALU wave-instructions per %s: %.1f (x 64 lanes / units);  FP64 arithmetic share of VALU instructions: %.0f %%"
          % (("leaves" if unit_name == "leaf" else unit_name + "s"), units, unit_name, valu * 64 / units, 100 * f64 / valu))
    print("#   waves %.0f, VALU instructions per wave %.0f, SALU per wave %.0f, LDS per wave %.0f"
          % (c["SQ_WAVES"], valu / c["SQ_WAVES"], c["SQ_INSTS_SALU"] / c["SQ_WAVES"], c["SQ_INSTS_LDS"] / c["SQ_WAVES"]))
    print("#   VALU busy: SQ_ACTIVE_INST_VALU x 4 / (%d SIMDs x GRBM_GUI_ACTIVE) = %.0f %%  (GRBM_GUI_ACTIVE / 8 XCDs = %.4g cycles = %.2f GHz over the kernel-trace duration)"
          % (SIMDS, 100 * c["SQ_ACTIVE_INST_VALU"] * 4 / (SIMDS * cycles), cycles, cycles / ns))
    print("#   cycles per VALU instruction while active: %.2f" % (c["SQ_ACTIVE_INST_VALU"] * 4 / valu))
    print("#   algorithmic: %.4g FLOP (%g per %s) / %.1f us = %.1f TFLOP/s = %.0f %% of the 78.6 TFLOP/s FP64 vector peak"
          % (alg, flop_per_unit, unit_name, ns / 1e3, alg / ns / 1e3, 100 * alg / (ns * 1e-9) / PEAK))
    print("#   hardware-counted FP64: %.4g FLOP (FMA = 2) = %.1f TFLOP/s = %.0f %% of peak" % (hw_flop, hw_flop / ns / 1e3, 100 * hw_flop / (ns * 1e-9) / PEAK))
    print("#   HBM: FETCH_SIZE %.4g KiB (x2 on gfx950) + WRITE_SIZE %.4g KiB = %.3g MB per launch" % (c["FETCH_SIZE"], c["WRITE_SIZE"], (2 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024 / 1e6))
    print()


if which == "tree":
    print("# rocprofv3 --pmc <one set per run> -- python3 tools/gpu_probe_tree_level.py fast|strict 7 2   (tools/collect_profiles_tree.sh; MI355X)")
    print("# The LRL-semicircle tree of test_controller.cpp:449-467 to depth 7.  Each node is simulate(r, t, 1.0) = 11 Euler substeps")
    print("# + one compareStates: 11 x 54 + 21 = 615 FLOP.  Counters: mean over the probe's launches with the kernel's largest grid.")
    print("# SQ_ACTIVE_INST_* / SQ_WAVE_CYCLES / SQ_WAIT_* count quad-cycles.")
    print("#")
    print("# <true> kernels = the largest STORED level, level 6 (17,827,584 nodes).  In a depth-7 search it is the level above the final")
    print("# one: only tiles that can matter to its own record or to the final level are simulated and stored, the others get 8 bytes")
    print("# per node (ghosts, ghost_select_kernel).  Simulated in full (MPC_HIP_TREE_GHOSTS=0) the same launch took 580 us (fast: 866")
    print("# VALU instructions per node, 69 % VALU-busy, 1.45 GB of stores) / 1497 us (strict: 2298 per node, 76 %, 1.87 GB).")
    print("# <false> kernels = the FINAL level (303,370,833 leaves), which is searched, not enumerated: the launch runs over the short")
    print("# list of tiles within reach of the checkpoint (tile_select_kernel) and skips waves that cannot win.")
    print("# For both, the 'algorithmic' lines price the reference's full enumeration of the level against the time the launch took --")
    print("# an equivalent rate, not arithmetic the GPU performed (the hardware-counted line is what it performed).")
    print()
    report("expand_fast_kernel<true> (MPC_PREC_FAST, stored level 6, with ghosts)", src + "/tree_fast", "expand_fast_kernel<true>", 17827584, "node", 615, True)
    report("expand_kernel<true> (MPC_PREC_F64, stored level 6, with ghosts)", src + "/tree_strict", "expand_kernel<true>", 17827584, "node", 615, True)
    report("expand_fast_kernel<false> (MPC_PREC_FAST, final level 7, short list)", src + "/tree_fast", "expand_fast_kernel<false>",
           303370833, "leaf", 615, True)
    report("expand_kernel<false> (MPC_PREC_F64, final level 7, short list)", src + "/tree_strict", "expand_kernel<false>",
           303370833, "leaf", 615, True)
else:
    print("# rocprofv3 --pmc <one set per run> -- python3 bench.py --precision f64 --no-pipeline --steps 20 --warmup 5 ...   (tools/collect_profiles_tree.sh; MI355X)")
    print("# rollout_kernel<MPC_PREC_F64, explicit, uniform>: the FP64 validation mode (BASELINE config 3's) on 65,536 candidates x 20 x 20:")
    print("# 1024 wavefronts = ONE per SIMD.  Unit = one Euler substep of one candidate (26,214,400 per launch); 54 FLOP each + 21 per score.")
    print()
    report("rollout_kernel<1, false, true, false, false>", src + "/strict", "rollout_kernel<1", 65536 * 400, "substep", (54 * 20 + 21) / 20.0, False)
