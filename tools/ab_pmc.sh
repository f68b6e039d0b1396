#!/bin/bash
# SQ_INSTS_VALU / SQ_ACTIVE_INST_VALU / SQ_WAVE_CYCLES of the config-2 pair kernel for the default build and the A/B build
set -e
export TMPDIR=/tmp
A="--steps 20 --warmup 5 --ramp-seconds 0 --no-cpu-baseline --no-tree --repeats 3"
for v in base ab; do
  if [ $v = ab ]; then export MPC_HIP_LIB=$PWD/mpc_amd/libmpc_hip_ab.so; else unset MPC_HIP_LIB; fi
  timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_SCA --output-format csv -d gpurun_out/abpmc_$v -- python3 bench.py $A > gpurun_out/abpmc_$v.log 2>&1
  echo == $v; python3 tools/pmc_summary.py gpurun_out/abpmc_$v | grep "rollout_group"
done
