#!/bin/bash
# Round profiles in one GPU call (run from the repo root on the GPU box): bench lines, rocprofv3 kernel stats of the
# bench command, PMC passes of the K1 kernels (bounded per pass), ablation and stamps of the forward pair kernel.
TAG=${1:-r03}
ROOT=$(pwd)
OUT="$ROOT/gpurun_out/$TAG"
mkdir -p "$OUT"
say() { echo "[collect] $*"; }
say "ablation library"; make -C gloria-nlp-project_amd/csrc ablate > "$OUT/make_ablate.log" 2>&1
say "bench default"; timeout -k 10 500 python bench.py > "$OUT/bench.json" 2> "$OUT/bench.err"
say "bench max lengths"; timeout -k 10 300 python bench.py --lengths max --no-cpu-baseline --steps 5 > "$OUT/bench_max_lengths.json" 2>> "$OUT/bench.err"
say "bench cfg2 fp32 B=64"; timeout -k 10 300 python bench.py --precision fp32 --global-batch 64 --no-cpu-baseline --steps 5 > "$OUT/bench_cfg2_fp32_b64.json" 2>> "$OUT/bench.err"
for gb in 128 64 32; do
  say "bench per-GPU batch $gb"; timeout -k 10 300 python bench.py --global-batch $gb --no-cpu-baseline > "$OUT/bench_b$gb.json" 2>> "$OUT/bench.err"
done
say "bench with the reference's training flags (submit_job.sh:15)"; timeout -k 10 300 python bench.py --train-flags --no-cpu-baseline > "$OUT/bench_train_flags.json" 2>> "$OUT/bench.err"
say "forced single-rank RCCL path at 32"
GLR_FORCE_DIST=1 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 MASTER_ADDR=127.0.0.1 MASTER_PORT=29571 timeout -k 10 300 python bench.py --global-batch 32 --no-cpu-baseline > "$OUT/bench_b32_forced_dist.json" 2>> "$OUT/bench.err"
say "rocprofv3 kernel stats of bench.py"
(cd /tmp && export TMPDIR=/tmp && timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof_bench" -- python3 "$ROOT/bench.py" --steps 5 --warmup 3 --no-cpu-baseline > "$OUT/bench_profiled.json" 2> "$OUT/prof_bench.err")
python tools/step_breakdown.py "$OUT"/prof_bench/*/*_kernel_trace.csv --steps 4 > "$OUT/step_breakdown.txt" 2>&1
cp "$OUT"/prof_bench/*/*_kernel_stats.csv "$OUT/bench_kernel_stats.csv" 2>/dev/null
rm -rf "$OUT/prof_bench"
say "rocprofv3 kernel stats of the loss forward + backward micro-benchmark"
(cd /tmp && export TMPDIR=/tmp && timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof_k1" -- python3 "$ROOT/tools/prof_k1.py" 256 bwd 6 > "$OUT/prof_k1.log" 2>&1)
cp "$OUT"/prof_k1/*/*_kernel_stats.csv "$OUT/k1_fwd_bwd_kernel_stats.csv" 2>/dev/null
rm -rf "$OUT/prof_k1"
say "PMC passes, K1 forward"; PASS_TIMEOUT=100 bash tools/pmc_k1.sh ${TAG}_fwd fwd > "$OUT/pmc_fwd.log" 2>&1
say "PMC passes, K1 forward + backward"; PASS_TIMEOUT=100 bash tools/pmc_k1.sh ${TAG}_bwd bwd > "$OUT/pmc_bwd.log" 2>&1
say "ablation"; timeout -k 10 300 python tools/ablate_k1_t1.py > "$OUT/ablate_t1.txt" 2>&1
say "K1 forward micro-benchmark (kernel-only / op-level, HIP events)"
{ timeout -k 10 120 python tools/bench_k1_kernel.py 256 30; GLR_K1_T1=0 timeout -k 10 120 python tools/bench_k1_kernel.py 256 30; timeout -k 10 120 python tools/bench_k1_kernel.py 256 10 max; NO_ATTN=1 timeout -k 10 120 python tools/bench_k1_kernel.py 256 30; } 2>&1 | grep "B=" > "$OUT/k1_microbench.txt"
say "ablation, backward"; timeout -k 10 200 python tools/ablate_k1_bwd.py > "$OUT/ablate_bwd.txt" 2>&1
say "encoder micro-benchmarks"; timeout -k 10 200 python tools/bench_bn.py > "$OUT/bench_bn.txt" 2>&1
(cd /tmp && export TMPDIR=/tmp && timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof_attn" -- python3 "$ROOT/tools/bench_attn.py" > "$OUT/bench_attn.txt" 2>&1)
grep -h "attn\|bwd_kernel" "$OUT"/prof_attn/*/*kernel_stats.csv > "$OUT/attn_kernel_stats.csv" 2>/dev/null
rm -rf "$OUT/prof_attn"
say "K3"; (cd /tmp && export TMPDIR=/tmp && timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof_k3" -- python3 "$ROOT/tools/bench_k3.py" > "$OUT/k3.log" 2>&1)
grep -h "k_global\|k_ce" "$OUT"/prof_k3/*/*kernel_stats.csv > "$OUT/k3_kernel_stats.csv" 2>/dev/null
rm -rf "$OUT/prof_k3"
say done
