#!/bin/bash
# PMC + kernel-trace passes for the kernels round 2 left without counter evidence (inside a gpurun call):
#   expand_fast_kernel<false> and expand_kernel<false> on the depth-7 level of the LRL-semicircle tree, and the strict
#   rollout kernel (rollout_kernel<1,...>: BASELINE config 3's mode) on the 65,536 x 20 x 20 set.
#     gpurun --timeout 900 -- 'bash tools/collect_profiles_tree.sh t1'   ->  gpurun_out/t1/...
# One counter set per run (8 SQ counters; FETCH_SIZE / WRITE_SIZE alone), the program directly after `--`.
set -e
D=gpurun_out/${1:-tree}
mkdir -p "$D"
export TMPDIR=/tmp
SQ_A="SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY"
SQ_B="GRBM_GUI_ACTIVE SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"
for mode in fast strict; do
    timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d "$D/tree_$mode/stats" -o t -- python3 tools/gpu_probe_tree_level.py $mode 7 3 > "$D/tree_$mode.log" 2>&1
    timeout -s KILL 120 rocprofv3 --pmc $SQ_A --output-format csv -d "$D/tree_$mode/sq_a" -- python3 tools/gpu_probe_tree_level.py $mode 7 2 > "$D/tree_$mode.sq_a.log" 2>&1
    timeout -s KILL 120 rocprofv3 --pmc $SQ_B --output-format csv -d "$D/tree_$mode/sq_b" -- python3 tools/gpu_probe_tree_level.py $mode 7 2 > "$D/tree_$mode.sq_b.log" 2>&1
    timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$D/tree_$mode/fetch" -- python3 tools/gpu_probe_tree_level.py $mode 7 2 > "$D/tree_$mode.fetch.log" 2>&1
    timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$D/tree_$mode/write" -- python3 tools/gpu_probe_tree_level.py $mode 7 2 > "$D/tree_$mode.write.log" 2>&1
done
F64="--precision f64 --no-pipeline --steps 20 --warmup 5 --ramp-seconds 0 --no-cpu-baseline --no-tree --repeats 3"
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d "$D/strict/stats" -o k -- python3 bench.py $F64 > "$D/strict.json" 2> "$D/strict.err"
timeout -s KILL 120 rocprofv3 --pmc $SQ_A --output-format csv -d "$D/strict/sq_a" -- python3 bench.py $F64 > "$D/strict.sq_a.log" 2>&1
timeout -s KILL 120 rocprofv3 --pmc $SQ_B --output-format csv -d "$D/strict/sq_b" -- python3 bench.py $F64 > "$D/strict.sq_b.log" 2>&1
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$D/strict/fetch" -- python3 bench.py $F64 > "$D/strict.fetch.log" 2>&1
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$D/strict/write" -- python3 bench.py $F64 > "$D/strict.write.log" 2>&1
echo done
