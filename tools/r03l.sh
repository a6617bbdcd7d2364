#!/bin/bash
OUT=gpurun_out/r03l; mkdir -p $OUT
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $OUT/gpu_tests.log 2>&1; tail -3 $OUT/gpu_tests.log | cut -c1-300
timeout -k 10 300 python bench.py --no-cpu-baseline --steps 20 > $OUT/bench.json 2> $OUT/bench.err
python3 - <<PY
import json
d=json.load(open("$OUT/bench.json")); r=d["roofline"]; l=d["loss_path"]
print("bench: %.2f ms/step  %.0f pairs/s  k1 %.3f ms (%.3f) op %.3f (%.3f)  bwd %.3f / op %.3f graph %s" % (d["ms_per_step"], d["value"], r["launch_ms"], r["frac"], r["op_ms"], r["frac_op"], l["k1_bwd_launch_ms"], l["k1_bwd_op_ms"], d["config"]["image_encoder_hipgraph"]))
PY
