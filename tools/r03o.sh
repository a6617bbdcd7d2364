#!/bin/bash
# fused-linear bias gradients: tests, then A/B bench
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_fused_ln.py -x -q -m gpu 2>&1 | tail -5 &&
for v in 1 0 1 0; do
  GLR_FUSED_LINEAR=$v timeout -k 10 400 python bench.py --no-cpu-baseline --steps 10 2>>gpurun_out/r03o.err | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('fused_linear=$v:', round(d['ms_per_step'],2), 'ms/step')"
done
