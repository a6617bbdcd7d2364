#!/bin/bash
# after the hipGraph fix: tests, then A/B benches (packet capture off)
mkdir -p gpurun_out
timeout -k 10 1100 python -m pytest tests/test_gpu_streams.py tests/test_gpu_train.py -x -q -m gpu 2>&1 | tail -8
E="GLR_FORCE_DIST=1 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 MASTER_ADDR=127.0.0.1"
b() { name=$1; gb=$2; steps=$3; shift 3; env "$@" timeout -k 10 400 python bench.py --no-cpu-baseline --global-batch $gb --steps $steps 2>>gpurun_out/r03r.err | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); c=d['config']; print('$name:', round(d['ms_per_step'],2), 'ms/step img_graph', c.get('image_encoder_hipgraph'), 'txt_graph', c.get('text_encoder_hipgraph'))"; }
b "b32 dist img+txt" 32 30 $E MASTER_PORT=29701
b "b32 dist img only" 32 30 $E MASTER_PORT=29702 GLR_GRAPH_TXT=0
b "b32 dist no graphs" 32 30 $E MASTER_PORT=29703 GLR_GRAPH_TXT=0 GLR_GRAPH_IMG=0
b "b32 dist img+txt (again)" 32 30 $E MASTER_PORT=29704
b "b64 dist img+txt" 64 20 $E MASTER_PORT=29705
b "b128 dist img+txt" 128 10 $E MASTER_PORT=29706
b "b256 single img+txt" 256 10
b "b256 single no graphs" 256 10 GLR_GRAPH_TXT=0 GLR_GRAPH_IMG=0
b "b32 single img+txt" 32 30
