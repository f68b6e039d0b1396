set -e
timeout -k 10 400 python -u -m pytest tests/test_gpu_pipeline.py tests/test_gpu_edges.py -x -q --timeout 300 --timeout-method thread > gpurun_out/r3_pipe.log 2>&1 || { tail -30 gpurun_out/r3_pipe.log; exit 1; }
tail -2 gpurun_out/r3_pipe.log
for L in 1 2 3 4; do
  timeout -k 10 120 python bench.py --no-cpu-baseline --no-tree --steps 24 --warmup 4 --pipeline-level $L > gpurun_out/lv_$L.json 2>gpurun_out/lv.err
  python -c "
import json; d=json.load(open('gpurun_out/lv_$L.json')); print($L, d['config']['solves_per_launch'], '%.4g'%d['value'], 'ms/step %.5f'%d['ms_per_step'], 'k1 %.5f'%d['rollout_kernel_ms'], 'frac %.3f'%d['roofline']['frac'])"
done
timeout -k 10 120 python bench.py --no-cpu-baseline --no-tree --steps 20 --warmup 5 > gpurun_out/lv_d.json 2>gpurun_out/lv.err
python -c "
import json; d=json.load(open('gpurun_out/lv_d.json')); print('driver-style', d['config']['solves_per_launch'], '%.4g'%d['value'], 'ms/step %.5f'%d['ms_per_step'], 'k1 %.5f'%d['rollout_kernel_ms'], 'frac %.3f'%d['roofline']['frac'])"
