#!/bin/bash
OUT=$(pwd)/gpurun_out/r03h; mkdir -p $OUT; ROOT=$(pwd)
export GLR_FORCE_DIST=1 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 MASTER_ADDR=127.0.0.1
for g in 0 1; do
(cd /tmp && export TMPDIR=/tmp && MASTER_PORT=2958$g GLR_GRAPH_IMG=$g timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof$g" -- python3 "$ROOT/bench.py" --global-batch 32 --steps 8 --warmup 3 --no-cpu-baseline > "$OUT/bench_prof$g.json" 2> "$OUT/prof$g.err")
python tools/step_breakdown.py "$OUT"/prof$g/*/*_kernel_trace.csv --steps 6 > "$OUT/step_breakdown_b32_dist_graph$g.txt" 2>&1
rm -rf "$OUT/prof$g"
head -14 "$OUT/step_breakdown_b32_dist_graph$g.txt"
done
unset GLR_FORCE_DIST RANK WORLD_SIZE LOCAL_RANK MASTER_ADDR
timeout -k 10 300 python -m pytest tests/test_gpu_streams.py -x -q 2>&1 | tail -3
