#!/bin/bash
# config-2 stream (driver-style regions) for several builds of the library on one box: base + mpc_amd/libmpc_hip_<name>.so
set -e
for r in 1 2; do
for v in base "$@"; do
  if [ $v = base ]; then unset MPC_HIP_LIB; else export MPC_HIP_LIB=$PWD/mpc_amd/libmpc_hip_$v.so; fi
  timeout -k 10 120 python bench.py --no-cpu-baseline --no-tree --steps 20 --warmup 5 > gpurun_out/var_$v.json 2>gpurun_out/var.err
  python -c "
import json; d=json.load(open('gpurun_out/var_$v.json')); print('$v', d['config']['solves_per_launch'], '%.4g'%d['value'], 'ms/step %.5f'%d['ms_per_step'], 'k1 %.5f'%d['rollout_kernel_ms'], 'frac %.3f'%d['roofline']['frac'], 'p50 %.4f'%d['p50_solve_latency_ms'])"
done
done
