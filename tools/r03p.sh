#!/bin/bash
# text-encoder hipGraph: tests, then A/B at 32 (forced dist) and 256 pairs
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_fused_ln.py tests/test_gpu_fused_attn.py tests/test_gpu_streams.py -x -q -m gpu 2>&1 | tail -15 &&
for v in 1 0 1 0; do
  GLR_GRAPH_TXT=$v GLR_FORCE_DIST=1 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 MASTER_ADDR=127.0.0.1 MASTER_PORT=2951$v timeout -k 10 400 python bench.py --no-cpu-baseline --steps 30 --global-batch 32 2>>gpurun_out/r03p.err | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('b32 dist graph_txt=$v:', round(d['ms_per_step'],2), 'ms/step', d['config'].get('text_encoder_hipgraph'))"
done
for v in 1 0; do
  GLR_GRAPH_TXT=$v timeout -k 10 400 python bench.py --no-cpu-baseline --steps 10 2>>gpurun_out/r03p.err | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('b256 graph_txt=$v:', round(d['ms_per_step'],2), 'ms/step', d['config'].get('text_encoder_hipgraph'))"
done
