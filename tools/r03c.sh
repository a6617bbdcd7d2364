#!/bin/bash
OUT=gpurun_out/r03c; mkdir -p $OUT
echo "[r03c] bench (MIOpen info log filtered)"
MIOPEN_LOG_LEVEL=5 timeout -k 10 420 python bench.py --no-cpu-baseline --steps 10 > $OUT/bench.json 2> $OUT/bench.err.full
grep -c "naive\|Naive" $OUT/bench.err.full > $OUT/naive_count.txt
grep "FindSolutionImpl" $OUT/bench.err.full | sort | uniq -c | sort -rn | head -20 > $OUT/find_solvers.txt
grep "\[bench\]" $OUT/bench.err.full > $OUT/bench.err; rm -f $OUT/bench.err.full
cat $OUT/bench.err; cut -c1-1200 $OUT/bench.json; cat $OUT/naive_count.txt; cat $OUT/find_solvers.txt
echo "[r03c] A/B old pair kernel"
GLR_K1_T1=0 timeout -k 10 420 python bench.py --no-cpu-baseline --steps 10 > $OUT/bench_pw.json 2> $OUT/bench_pw.err; cut -c1-300 $OUT/bench_pw.json
echo "[r03c] new tests"
timeout -k 10 600 python -m pytest tests/test_gpu_streams.py tests/test_gpu_train.py tests/test_gpu_aux.py tests/test_gpu_parity.py -x -q > $OUT/tests.log 2>&1; tail -15 $OUT/tests.log
