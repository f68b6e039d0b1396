#!/bin/bash
# depth-7 fast tree search for several builds of the library on one box (timing only: ablation builds give wrong answers)
for v in base "$@"; do
  if [ $v = base ]; then unset MPC_HIP_LIB; else export MPC_HIP_LIB=$PWD/mpc_amd/libmpc_hip_$v.so; fi
  echo "$v $(timeout -k 10 120 python tools/gpu_probe_tree_level.py fast 7 4 2>&1 | tail -2 | sed 's/.*gpu_ms=\([0-9.]*\).*/\1/' | tr '\n' ' ')"
done
