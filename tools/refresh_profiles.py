"""Turn one profiling run under gpurun_out/<dir> (stats/, FETCH_SIZE/, WRITE_SIZE/, SQ_INSTS_VALU/, GRBM_GUI_ACTIVE/,
bench_prof.log, optional tree/) into the committed summaries profiles/<tag>_*.  Usage:
    python tools/refresh_profiles.py gpurun_out/p2 r01
"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

src, tag = sys.argv[1], sys.argv[2]
out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles")


def stats_table(path, title, tail=()):
    rows = list(csv.DictReader(open(path)))
    lines = ["# " + title, "# name | calls | total_ns | avg_ns | pct | min_ns | max_ns"]
    for r in rows:
        lines.append(" | ".join([r["Name"][:60], r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"],
                                 r["MinNs"], r["MaxNs"]]))
    lines += list(tail)
    return "\n".join(lines) + "\n", rows


bench_line = None
for ln in open(os.path.join(src, "bench_prof.log")):
    if ln.startswith("{"):
        bench_line = json.loads(ln)
tail = []
if bench_line:
    tail = ["", "# bench.py JSON line of the same (profiled) run: rollout_kernel_ms=%.5f value=%.4e steps/s roofline.frac=%.4f"
            % (bench_line["rollout_kernel_ms"], bench_line["value"], bench_line["roofline"]["frac"])]
txt, rows = stats_table(os.path.join(src, "stats", "r1_kernel_stats.csv"),
                        "rocprofv3 --kernel-trace --stats --output-format csv -- python3 bench.py"
                        "   (MI355X, fast precision, config 2, default steps/warmup)", tail)
open(os.path.join(out, tag + "_bench_kernel_stats.txt"), "w").write(txt)
with open(os.path.join(out, tag + "_bench_kernel_stats.csv"), "w") as f:
    f.write(open(os.path.join(src, "stats", "r1_kernel_stats.csv")).read())

for sub, arg, what in (("tree", "", "strict precision"), ("treefast", " fast", "fast expansion, strict decision")):
    tree = os.path.join(src, sub, "t_kernel_stats.csv")
    if os.path.exists(tree):
        extra = ["", "# tools/gpu_probe_tree.py output of the same run:"]
        extra += ["# " + ln.rstrip() for ln in open(os.path.join(src, sub + ".log")) if "depth=" in ln]
        txt, _ = stats_table(tree, "rocprofv3 --kernel-trace --stats --output-format csv -- python3 tools/gpu_probe_tree.py" + arg +
                             "   (tree search, depths 3..7, " + what + ")", extra)
        open(os.path.join(out, tag + "_" + sub + "_kernel_stats.txt"), "w").write(txt)

acc = defaultdict(lambda: [0.0, 0])
for path in glob.glob(src + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(path)):
        k = (r["Kernel_Name"].split("(")[0][:48], r["Counter_Name"])
        acc[k][0] += float(r["Counter_Value"])
        acc[k][1] += 1
lines = ["# rocprofv3 --pmc <set> --output-format csv -- python3 bench.py --steps 20 --warmup 5 --ramp-seconds 0 --no-cpu-baseline",
         "# four separate passes (FETCH_SIZE | WRITE_SIZE | SQ_* | GRBM_GUI_ACTIVE TCC_*), MI355X, mean per launch",
         "# FETCH_SIZE / WRITE_SIZE are in KiB; gfx950 FETCH_SIZE counts 64 B per 128-B request (MI355X_MICROARCH.md, HBM): double it",
         "# SQ_WAVE_CYCLES / SQ_ACTIVE_INST_* / SQ_WAIT_* count quad-cycles"]
k1 = {}
# the timed region's kernel: rollout_group_kernel when the bench ran at pipeline level >= 2 (several solves per launch;
# the latency loop at the end of bench.py launches plain rollout_kernel + finish_kernel, one solve at a time)
paired = any("rollout_group_kernel" in kern for (kern, _c) in acc)
spl = int(bench_line["config"]["solves_per_launch"]) if bench_line else (4 if paired else 1)
for (kern, ctr), (s, n) in sorted(acc.items()):
    if "rollout" in kern or "finish" in kern:
        lines.append("%-50s %-24s launches=%-5d mean=%.6g" % (kern, ctr, n, s / n))
    if ("rollout_group_kernel" in kern) if paired else ("rollout" in kern):
        k1[ctr] = s / n
open(os.path.join(out, tag + "_pmc_summary.txt"), "w").write("\n".join(lines) + "\n")
summary = {
    "kernel": ("rollout_group_kernel<explicit>: %d solves per launch + the finish workgroups of the solves of the launch before" % spl if paired else
               "rollout_kernel<fast, explicit, uniform schedule, fused finish of the previous solve>"),
    "solves_per_launch": spl if paired else 1, "workload": "config 2 (65536 x 20 x 20), a stream of distinct candidate sets",
    "fetch_size_kib": k1["FETCH_SIZE"], "write_size_kib": k1["WRITE_SIZE"],
    "hbm_bytes_per_launch": int(round((2 * k1["FETCH_SIZE"] + k1["WRITE_SIZE"]) * 1024)),
    "note": "hbm_bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024: FETCH_SIZE doubled per the gfx950 correction; 21 MB of the "
            "writes are the state snapshots of every 4th level (65536 x 5 slots x 4 pairs x 16 B) that replace the winner's "
            "second rollout (round 1 wrote every level: 83.9 MB), per solve; the launch also carries the finish workgroups of the solves before",
    "sq_insts_valu": k1["SQ_INSTS_VALU"], "sq_waves": k1["SQ_WAVES"], "sq_wave_cycles_quad": k1["SQ_WAVE_CYCLES"],
    "sq_active_inst_valu_quad": k1["SQ_ACTIVE_INST_VALU"], "sq_wait_any_quad": k1.get("SQ_WAIT_ANY"),
    "grbm_gui_active": k1["GRBM_GUI_ACTIVE"],
}
json.dump(summary, open(os.path.join(out, tag + "_pmc_summary.json"), "w"), indent=1)
print(json.dumps(summary))
