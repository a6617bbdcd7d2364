#!/bin/bash
OUT=gpurun_out/r03f; mkdir -p $OUT; rm -f $OUT/k1_micro.txt
for v in "" sched "" sched; do GLR_LIB_VARIANT=$v timeout -k 10 120 python tools/bench_k1_kernel.py 256 30 2>&1 | grep "B=" | sed "s/^/variant [$v]: /" >> $OUT/k1_micro.txt; done
cat $OUT/k1_micro.txt
