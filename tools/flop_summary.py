"""rocprof-counted FP64 operations of the rollout kernel -> achieved hardware FLOP/s (north_star: "rocprof-reported
achieved VALU FLOP/s ... against MI355X peak").  Input: gpurun_out/<dir> of tools/collect_profiles_dense.sh plus the
config-2 pass of the same counters; kernel durations from the kernel-trace stats of the same directory (config 2: from
profiles/<tag>_bench_kernel_stats.csv).  Usage: python tools/flop_summary.py gpurun_out/d1 r01 > profiles/r01_fp64_flop_counters.txt"""
import csv
import glob
import os
import sys
from collections import defaultdict

src, tag = sys.argv[1], sys.argv[2]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PEAK = 78.6e12


def counters(d, kernel="rollout_kernel"):
    acc = defaultdict(lambda: [0.0, 0])
    for path in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(path)):
            if kernel in r["Kernel_Name"]:
                acc[r["Counter_Name"]][0] += float(r["Counter_Value"]); acc[r["Counter_Name"]][1] += 1
    return {k: s / n for k, (s, n) in acc.items()}


def k1_avg_ns(stats_csv, kernel="rollout_kernel"):
    for r in csv.DictReader(open(stats_csv)):
        if kernel in r["Name"]:
            return float(r["AverageNs"])
    return None


print("# rocprofv3 --pmc SQ_INSTS_VALU_{FMA,MUL,ADD,TRANS}_F64 (wave-instructions, mean per launch of rollout_kernel) x 64 lanes,")
print("# FMA counted as 2 FLOP; duration = rocprofv3 --kernel-trace average of the same workload; peak 78.6 TFLOP/s FP64 vector")
print("# config2 = one solve per launch (rollout_kernel), config2_group = four solves per launch (rollout_group_kernel)")
print("# workload | fma | mul | add | trans | FP64 FLOP per launch | K1 avg us | achieved TFLOP/s | fraction of peak")
# config2: bench.py's timed region launches rollout_group_kernel (four solves per launch, pipeline level 4), its latency
# loop rollout_kernel (one solve per launch)
c2 = os.path.join(root, "profiles", tag + "_bench_kernel_stats.csv")
for w, d, stats, kernel in (("config2", "config2", c2, "rollout_kernel"),
                            ("config2_group", "config2", c2, "rollout_group_kernel"),
                            ("config4", "config4", os.path.join(src, "config4", "stats", "k_kernel_stats.csv"), "rollout_kernel"),
                            ("config5", "config5", os.path.join(src, "config5", "stats", "k_kernel_stats.csv"), "rollout_kernel")):
    c = counters(os.path.join(src, d, "flops"), kernel)
    ns = k1_avg_ns(stats, kernel)
    if not c or ns is None:
        continue
    fma, mul, add, tr = (c["SQ_INSTS_VALU_FMA_F64"], c["SQ_INSTS_VALU_MUL_F64"], c["SQ_INSTS_VALU_ADD_F64"], c["SQ_INSTS_VALU_TRANS_F64"])
    flop = (2 * fma + mul + add + tr) * 64
    print("%s | %.4g | %.4g | %.4g | %.4g | %.4g | %.1f | %.1f | %.3f" % (w, fma, mul, add, tr, flop, ns / 1e3, flop / ns / 1e3, flop / (ns * 1e-9) / PEAK))
