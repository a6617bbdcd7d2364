#!/bin/bash
# PMC counters of the collate kernels (separate rocprofv3 --pmc passes, no tracing); run on the GPU box from the repo root.
set -e
ROOT=$(pwd)
OUT="$ROOT/gpurun_out/pmc_collate"
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU" \
           "SQ_INSTS_LDS SQ_INSTS_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS" \
           "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS" \
           "FETCH_SIZE WRITE_SIZE TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i + 1))
  rocprofv3 --pmc $set --output-format csv -d "$OUT/pass$i" -- python3 "$ROOT/tools/bench_collate.py" 16 3056 2544 2 > "$OUT/pass$i.log" 2>&1
  echo "pass $i done"
done
python3 "$ROOT/tools/pmc_summary.py" "$OUT" k_collate > "$ROOT/gpurun_out/collate_pmc.txt"
python3 "$ROOT/tools/pmc_summary.py" "$OUT" k_image_minmax >> "$ROOT/gpurun_out/collate_pmc.txt"
cat "$ROOT/gpurun_out/collate_pmc.txt"
