#!/bin/bash
# rocprofv3 kernel stats of the collate micro-benchmark; run on the GPU box from the repo root.
set -e
ROOT=$(pwd)
mkdir -p "$ROOT/gpurun_out/prof_collate"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$ROOT/gpurun_out/prof_collate" -- python3 "$ROOT/tools/bench_collate.py" 64 > "$ROOT/gpurun_out/prof_collate/bench.log" 2>&1
find "$ROOT/gpurun_out/prof_collate" -name "*kernel_stats.csv" -exec cp {} "$ROOT/gpurun_out/collate_kernel_stats.csv" \;
tail -2 "$ROOT/gpurun_out/prof_collate/bench.log"
