#!/bin/bash
# PMC counters of the collate kernels (separate rocprofv3 --pmc passes, no tracing); run on the GPU box from the repo root.
# Counter sets are sized to the per-pass slots of gfx950 (MI355X_MICROARCH.md "rocprofv3 PMC slots": SQ 8, TCC 4 with
# FETCH_SIZE costing 3 and WRITE_SIZE 2): round 1 asked for FETCH_SIZE + WRITE_SIZE + two TCC counters in ONE pass
# (7 of 4 TCC slots), rocprofv3 aborted with "error code 38: Request exceeds the capabilities of the hardware" and then
# sat in finalisation until the call's limit.  Hence FETCH_SIZE alone, and a per-pass timeout on every profiler run.
set -e
ROOT=$(pwd)
OUT="$ROOT/gpurun_out/pmc_collate"
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
PASS_TIMEOUT=${PASS_TIMEOUT:-150}
i=0
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU" \
           "SQ_INSTS_LDS SQ_INSTS_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS" \
           "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS" \
           "FETCH_SIZE" \
           "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i + 1))
  timeout -k 10 "$PASS_TIMEOUT" rocprofv3 --pmc $set --output-format csv -d "$OUT/pass$i" -- python3 "$ROOT/tools/bench_collate.py" 16 3056 2544 2 > "$OUT/pass$i.log" 2>&1 \
    || { echo "pass $i ($set) failed or timed out: see $OUT/pass$i.log"; tail -5 "$OUT/pass$i.log"; exit 1; }
  echo "pass $i done"
done
python3 "$ROOT/tools/pmc_summary.py" "$OUT" k_collate > "$ROOT/gpurun_out/collate_pmc.txt"
python3 "$ROOT/tools/pmc_summary.py" "$OUT" k_image_minmax >> "$ROOT/gpurun_out/collate_pmc.txt"
cat "$ROOT/gpurun_out/collate_pmc.txt"
