#!/bin/bash
# Profiles of the dense workloads (configs 4 and 5: >= 32 waves per SIMD over the launch), inside a gpurun call:
#     gpurun --timeout 900 -- 'bash tools/collect_profiles_dense.sh d1'
# -> gpurun_out/<dir>/{config4,config5}/...; summarise with tools/pmc_summary.py.
set -e
D=gpurun_out/${1:-dense}
mkdir -p "$D"
export TMPDIR=/tmp
# the same FP64-operation counters for config 2 (tools/flop_summary.py)
mkdir -p "$D/config2"
timeout -s KILL 120 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_TRANS_F64 \
    --output-format csv -d "$D/config2/flops" -- python3 bench.py --steps 20 --warmup 5 --ramp-seconds 0 --no-cpu-baseline --no-tree > "$D/config2.flops.log" 2>&1
for w in config4 config5; do
    timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$D/$w/stats" -o k -- python3 bench.py --workload $w --steps 20 --warmup 3 > "$D/$w.json" 2> "$D/$w.err"
    timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY \
        --output-format csv -d "$D/$w/sq" -- python3 bench.py --workload $w --steps 5 --warmup 2 > "$D/$w.sq.log" 2>&1
    timeout -s KILL 120 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_TRANS_F64 \
        --output-format csv -d "$D/$w/flops" -- python3 bench.py --workload $w --steps 5 --warmup 2 > "$D/$w.flops.log" 2>&1
done
echo done
