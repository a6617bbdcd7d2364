#!/bin/bash
# PMC counters of the K1 forward kernels (separate rocprofv3 --pmc passes, each bounded by `timeout`; FETCH_SIZE in a
# pass of its own: it takes 3 of the 4 TCC slots).  Run on the GPU box from the repo root:
#   bash tools/pmc_k1.sh <tag> [fwd|bwd]          GLR_K1_PAIR_V1=1 in the environment profiles the round-1 pair kernel
set -e
TAG=${1:-k1}
MODE=${2:-fwd}
ROOT=$(pwd)
OUT="$ROOT/gpurun_out/pmc_$TAG"
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
PASS_TIMEOUT=${PASS_TIMEOUT:-120}
ARGS="256"
[ "$MODE" = "bwd" ] && ARGS="256 bwd 3"
i=0
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_VALU_MFMA_BUSY_CYCLES" \
           "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
           "FETCH_SIZE" \
           "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i + 1))
  [ -n "$PMC_PASSES" ] && [[ ! " $PMC_PASSES " =~ " $i " ]] && continue
  timeout -k 10 "$PASS_TIMEOUT" rocprofv3 --pmc $set --output-format csv -d "$OUT/pass$i" -- python3 "$ROOT/tools/prof_k1.py" $ARGS > "$OUT/pass$i.log" 2>&1 \
    || { echo "pass $i ($set) failed or timed out"; tail -5 "$OUT/pass$i.log"; exit 1; }
done
python3 "$ROOT/tools/pmc_summary.py" "$OUT" k_local_attn > "$ROOT/gpurun_out/pmc_$TAG.txt"
cat "$ROOT/gpurun_out/pmc_$TAG.txt"
