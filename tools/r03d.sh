#!/bin/bash
OUT=gpurun_out/r03d; mkdir -p $OUT
echo "[r03d] kernel-only microbench"
for ib in 2 4 8; do GLR_K1_IMG_BLOCK=$ib timeout -k 10 120 python tools/bench_k1_kernel.py 256 20 >> $OUT/k1_micro.txt 2>&1; done
GLR_K1_T1=0 timeout -k 10 120 python tools/bench_k1_kernel.py 256 20 >> $OUT/k1_micro.txt 2>&1
timeout -k 10 120 python tools/bench_k1_kernel.py 256 10 max >> $OUT/k1_micro.txt 2>&1
cat $OUT/k1_micro.txt
echo "[r03d] ablation"
timeout -k 10 300 python tools/ablate_k1_t1.py > $OUT/ablate_t1.txt 2>&1; cat $OUT/ablate_t1.txt
echo "[r03d] PMC"
PASS_TIMEOUT=100 bash tools/pmc_k1.sh r03d fwd > $OUT/pmc.log 2>&1; tail -50 gpurun_out/pmc_r03d.txt
echo "[r03d] kernel stats of the forward op"
(cd /tmp && export TMPDIR=/tmp && timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d "$OLDPWD/$OUT/prof_k1" -- python3 "$OLDPWD/tools/bench_k1_kernel.py" 256 10 > "$OLDPWD/$OUT/prof_k1.log" 2>&1)
cp $OUT/prof_k1/*/*_kernel_stats.csv $OUT/k1_fwd_kernel_stats.csv 2>/dev/null; rm -rf $OUT/prof_k1
cut -d, -f1-4 $OUT/k1_fwd_kernel_stats.csv | cut -c1-160 | head -20
echo "[r03d] stream tests + bench (perf-db back, naive off)"
timeout -k 10 300 python -m pytest tests/test_gpu_streams.py -x -q > $OUT/tests.log 2>&1; tail -5 $OUT/tests.log
timeout -k 10 400 python bench.py --no-cpu-baseline --steps 10 > $OUT/bench.json 2> $OUT/bench.err; grep bench $OUT/bench.err; cut -c1-900 $OUT/bench.json
