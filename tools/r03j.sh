#!/bin/bash
OUT=gpurun_out/r03j; mkdir -p $OUT
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q > $OUT/tests.log 2>&1; tail -2 $OUT/tests.log
for i in 1 2; do timeout -k 10 120 python tools/bench_k1_kernel.py 256 30 2>&1 | grep "B="; done
timeout -k 10 300 python tools/ablate_k1_t1.py 2>&1 | grep -v amdgpu.ids > $OUT/ablate_t1.txt; cat $OUT/ablate_t1.txt
