#!/bin/bash
# Round-3 evidence run with the final kernels (inside a gpurun call): bench lines + the long random runs against the oracle.
set -e
mkdir -p gpurun_out/ev3
timeout -k 10 400 python bench.py > gpurun_out/ev3/bench_default.json 2> gpurun_out/ev3/bench_default.err
timeout -k 10 400 python bench.py --steps 20 --warmup 5 > gpurun_out/ev3/bench_driver.json 2> gpurun_out/ev3/bench_driver.err
timeout -k 10 400 python bench.py --gpus 2 --backend gloo --same-device --steps 20 --warmup 5 --combine-batch 8 > gpurun_out/ev3/bench_gloo2.json 2> gpurun_out/ev3/bench_gloo2.err
timeout -k 10 600 python -u tools/fuzz_tree.py 3000 12 > gpurun_out/ev3/fuzz_tree_12.txt 2>&1
timeout -k 10 600 python -u tools/fuzz_tree.py 3000 13 > gpurun_out/ev3/fuzz_tree_13.txt 2>&1
timeout -k 10 600 python -u tools/fuzz_random_problems.py 100 100 > gpurun_out/ev3/fuzz_random.txt 2>&1
timeout -k 10 600 python -u tools/fuzz_explicit.py 1500 3 > gpurun_out/ev3/fuzz_explicit.txt 2>&1
timeout -k 10 600 python -u tools/fuzz_shards.py 600 9 > gpurun_out/ev3/fuzz_shards.txt 2>&1
timeout -k 10 600 python -u tools/fuzz_fleet.py > gpurun_out/ev3/fuzz_fleet.txt 2>&1
echo done
