#!/bin/bash
OUT=gpurun_out/r03i; mkdir -p $OUT
timeout -k 10 600 python -m pytest tests/test_gpu_aux.py tests/test_gpu_parity.py -x -q > $OUT/tests.log 2>&1; tail -3 $OUT/tests.log
for gk in 1 0; do GLR_GRAM_KERNEL=$gk timeout -k 10 120 python tools/bench_k1_kernel.py 256 30 2>&1 | grep "B=" | sed "s/^/gram kernel $gk: /"; done
timeout -k 10 300 python bench.py --no-cpu-baseline --steps 10 > $OUT/bench.json 2> $OUT/bench.err
python3 - <<PY
import json
d=json.load(open("$OUT/bench.json")); r=d["roofline"]; l=d["loss_path"]
print("bench: %.2f ms/step  %.0f pairs/s  k1 %.3f ms (%.3f) op %.3f (%.3f)  bwd %.3f / op %.3f" % (d["ms_per_step"], d["value"], r["launch_ms"], r["frac"], r["op_ms"], r["frac_op"], l["k1_bwd_launch_ms"], l["k1_bwd_op_ms"]))
PY
