#!/bin/bash
# One profiling run on the GPU box (inside a gpurun call), everything under gpurun_out/<dir>:
#     gpurun --timeout 900 -- 'bash tools/collect_profiles.sh p7'
# then here:  python tools/refresh_profiles.py gpurun_out/p7 r01   ->  profiles/r01_*
# Each GPU step has its own time limit and the script stops at the first failure (no retries).
set -e
D=gpurun_out/${1:-prof}
mkdir -p "$D"
export TMPDIR=/tmp
PMC_ARGS="--steps 20 --warmup 5 --ramp-seconds 0 --no-cpu-baseline --no-tree"

# kernel trace of the default bench command (the numbers roofline.achieved is checked against)
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$D/stats" -o r1 -- python3 bench.py > "$D/bench_prof.log" 2> "$D/bench_prof.err"
# counters: one pass per set (the SQ / TCC blocks have 8 / 4 counters; FETCH_SIZE takes 3 TCC counters, WRITE_SIZE 2)
timeout -s KILL 90 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$D/FETCH_SIZE" -- python3 bench.py $PMC_ARGS > "$D/FETCH_SIZE.log" 2>&1
timeout -s KILL 90 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$D/WRITE_SIZE" -- python3 bench.py $PMC_ARGS > "$D/WRITE_SIZE.log" 2>&1
timeout -s KILL 90 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY \
    --output-format csv -d "$D/SQ_INSTS_VALU" -- python3 bench.py $PMC_ARGS > "$D/SQ_INSTS_VALU.log" 2>&1
timeout -s KILL 90 rocprofv3 --pmc GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum --output-format csv -d "$D/GRBM_GUI_ACTIVE" -- python3 bench.py $PMC_ARGS > "$D/GRBM_GUI_ACTIVE.log" 2>&1
# tree search, both expansions
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d "$D/tree" -o t -- python3 tools/gpu_probe_tree.py > "$D/tree.log" 2>&1
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d "$D/treefast" -o t -- python3 tools/gpu_probe_tree.py fast > "$D/treefast.log" 2>&1
# unprofiled numbers
timeout -k 10 300 python3 bench.py > "$D/bench_plain.json" 2> "$D/bench_plain.err"
timeout -k 10 300 python3 bench.py --workload config4 > "$D/bench_config4.json" 2> "$D/bench_config4.err"
timeout -k 10 300 python3 bench.py --workload config5 > "$D/bench_config5.json" 2> "$D/bench_config5.err"
timeout -k 10 300 python3 bench.py --precision f64 --steps 300 --warmup 30 > "$D/bench_f64.json" 2> "$D/bench_f64.err"
timeout -k 10 200 python3 tools/gpu_probe_tree.py > "$D/tree_plain.log" 2>&1
timeout -k 10 200 python3 tools/gpu_probe_tree.py fast > "$D/treefast_plain.log" 2>&1
timeout -k 10 300 python3 tools/gpu_probe.py > "$D/probe.log" 2>&1
timeout -k 10 200 python3 tools/concurrent_solves.py > "$D/conc.log" 2>&1
echo done
