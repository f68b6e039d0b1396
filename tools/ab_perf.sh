#!/bin/bash
# A/B of two builds of the library on ONE box in ONE call (boxes differ by ~3 %): the default build against
# mpc_amd/libmpc_hip_ab.so (MPC_HIP_LIB), alternating, config 2 / 4 / 5 and the depth-7 tree.
#     gpurun -- 'bash tools/ab_perf.sh [rounds]'
set -e
B="--no-cpu-baseline --no-tree"
R=${1:-2}
for r in $(seq 1 $R); do
  for v in base ab; do
    if [ $v = ab ]; then export MPC_HIP_LIB=$PWD/mpc_amd/libmpc_hip_ab.so; else unset MPC_HIP_LIB; fi
    timeout -k 10 120 python bench.py $B --steps 20 --warmup 5 > gpurun_out/ab_${v}_c2_$r.json 2>gpurun_out/ab.err
    timeout -k 10 120 python bench.py $B --workload config4 > gpurun_out/ab_${v}_c4_$r.json 2>gpurun_out/ab.err
    timeout -k 10 120 python bench.py $B --workload config5 > gpurun_out/ab_${v}_c5_$r.json 2>gpurun_out/ab.err
    timeout -k 10 120 python tools/gpu_probe_tree_level.py fast 7 4 > gpurun_out/ab_${v}_tf_$r.log 2>&1
  done
done
unset MPC_HIP_LIB
python - <<PY
import json
for r in range(1, $R + 1):
    for v in ("base", "ab"):
        row = [v, r]
        for w in ("c2", "c4", "c5"):
            d = json.load(open("gpurun_out/ab_%s_%s_%d.json" % (v, w, r)))
            row.append("%s %.4g" % (w, d["value"]))
            if w == "c2":
                row.append("1/launch %.4g p50 %.4f" % (d["one_solve_per_launch"]["value"], d["p50_solve_latency_ms"]))
        row.append("tree7 " + open("gpurun_out/ab_%s_tf_%d.log" % (v, r)).read().strip().splitlines()[-1].split("gpu_ms=")[1].split()[0])
        print(*row)
PY
