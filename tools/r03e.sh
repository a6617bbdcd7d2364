#!/bin/bash
OUT=gpurun_out/r03e; mkdir -p $OUT
echo "[r03e] parity tests"
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_aux.py -x -q > $OUT/tests.log 2>&1; tail -4 $OUT/tests.log
echo "[r03e] kernel-only microbench"
for ib in 2 4; do GLR_K1_IMG_BLOCK=$ib timeout -k 10 120 python tools/bench_k1_kernel.py 256 20 2>&1 | grep "B=" >> $OUT/k1_micro.txt; done
GLR_K1_T1=0 timeout -k 10 120 python tools/bench_k1_kernel.py 256 20 2>&1 | grep "B=" >> $OUT/k1_micro.txt
timeout -k 10 120 python tools/bench_k1_kernel.py 256 10 max 2>&1 | grep "B=" >> $OUT/k1_micro.txt
cat $OUT/k1_micro.txt
echo "[r03e] ablation"
timeout -k 10 300 python tools/ablate_k1_t1.py 2>&1 | grep -v amdgpu.ids > $OUT/ablate_t1.txt; cat $OUT/ablate_t1.txt
