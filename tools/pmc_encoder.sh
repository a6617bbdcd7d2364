#!/bin/bash
# PMC counters (HBM traffic) of the fused encoder kernels: separate, bounded rocprofv3 --pmc passes over the micro-benchmarks.
#   bash tools/pmc_encoder.sh <tag>      (GPU box, repo root)
TAG=${1:-enc}
ROOT=$(pwd)
OUT="$ROOT/gpurun_out/pmc_$TAG"
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
PASS_TIMEOUT=${PASS_TIMEOUT:-120}
i=0
for set in "FETCH_SIZE" "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i + 1))
  B=64 timeout -k 10 "$PASS_TIMEOUT" rocprofv3 --pmc $set --output-format csv -d "$OUT/bn_pass$i" -- python3 "$ROOT/tools/bench_bn.py" --only-fused --shapes big --batch 64 --iters 2 > "$OUT/bn_pass$i.log" 2>&1 || { echo "bn pass $i failed"; tail -3 "$OUT/bn_pass$i.log"; }
  B=64 timeout -k 10 "$PASS_TIMEOUT" rocprofv3 --pmc $set --output-format csv -d "$OUT/attn_pass$i" -- python3 "$ROOT/tools/bench_attn.py" > "$OUT/attn_pass$i.log" 2>&1 || { echo "attn pass $i failed"; tail -3 "$OUT/attn_pass$i.log"; }
done
{ echo "# fused BatchNorm kernels, (64, 64, 150, 150) and (64, 256, 75, 75) bf16: E = 184 MB per tensor; FETCH_SIZE / WRITE_SIZE in KB, FETCH x2 on gfx950";
  python3 "$ROOT/tools/pmc_summary.py" "$OUT" k_bn_;
  echo "# self-attention kernels, B = 64 x 12 heads x 97 tokens: q, k, v, o, dO = 9.5 MB each";
  python3 "$ROOT/tools/pmc_summary.py" "$OUT" k_attn_; } > "$ROOT/gpurun_out/pmc_$TAG.txt"
cat "$ROOT/gpurun_out/pmc_$TAG.txt"
