#!/bin/bash
# SQ counters of the depth-7 fast expansion for the default build and A/B builds (mpc_amd/libmpc_hip_<name>.so)
export TMPDIR=/tmp
for v in base "$@"; do
  if [ $v = base ]; then unset MPC_HIP_LIB; else export MPC_HIP_LIB=$PWD/mpc_amd/libmpc_hip_$v.so; fi
  timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_WAIT_INST_ANY --output-format csv -d gpurun_out/tpmc_$v -- python3 tools/gpu_probe_tree_level.py fast 7 2 > gpurun_out/tpmc_$v.log 2>&1
  echo == $v; python3 tools/pmc_summary.py gpurun_out/tpmc_$v | grep "expand_fast_kernel<false>"
done
