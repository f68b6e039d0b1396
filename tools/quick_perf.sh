set -e
B="--no-cpu-baseline --no-tree"
timeout -k 10 120 python bench.py $B --steps 20 --warmup 5 > gpurun_out/q_c2.json 2>gpurun_out/q_c2.err
timeout -k 10 120 python bench.py $B --workload config4 > gpurun_out/q_c4.json 2>gpurun_out/q_c4.err
timeout -k 10 120 python bench.py $B --workload config5 > gpurun_out/q_c5.json 2>gpurun_out/q_c5.err
timeout -k 10 120 python tools/gpu_probe_tree_level.py fast 7 4 > gpurun_out/q_tf.log 2>&1
timeout -k 10 120 python tools/gpu_probe_tree_level.py strict 7 3 > gpurun_out/q_ts.log 2>&1
python - <<'PY'
import json
for w in ("c2","c4","c5"):
    d=json.load(open("gpurun_out/q_%s.json"%w))
    print(w, "%.4g"%d["value"], "ms/step %.5f"%d["ms_per_step"], "k1 %.5f"%d["rollout_kernel_ms"], "frac %.3f"%d["roofline"]["frac"], "one/launch", (d.get("one_solve_per_launch") or {}).get("value"), "p50", d.get("p50_solve_latency_ms"))
PY
tail -2 gpurun_out/q_tf.log; tail -1 gpurun_out/q_ts.log
