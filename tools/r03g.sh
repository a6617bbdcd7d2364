#!/bin/bash
OUT=gpurun_out/r03g; mkdir -p $OUT
echo "[r03g] full GPU test suite"
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $OUT/gpu_tests.log 2>&1; tail -3 $OUT/gpu_tests.log
run() { name=$1; shift; timeout -k 10 300 env "$@" python bench.py --no-cpu-baseline --steps 10 ${EXTRA} > $OUT/$name.json 2>> $OUT/bench.err; python3 - <<PY
import json
try:
    d=json.load(open("$OUT/$name.json")); r=d["roofline"]; l=d["loss_path"]
    print("$name: %.2f ms/step  %.0f pairs/s  first_step %.1fs  k1 %.3f ms (%.3f) op %.3f (%.3f)  bwd %.3f / op %.3f  launches %s" % (d["ms_per_step"], d["value"], d["config"]["first_step_s"], r["launch_ms"], r["frac"], r["op_ms"], r["frac_op"], l["k1_bwd_launch_ms"], l["k1_bwd_op_ms"], d["config"]["kernel_launches_per_step"]))
except Exception as e: print("$name failed", e)
PY
}
EXTRA="" run b256 A=1
EXTRA="" run b256_graph GLR_GRAPH_IMG=1
EXTRA="--train-flags" run b256_trainflags A=1
for gb in 128 64 32; do
  EXTRA="--global-batch $gb" run b${gb} A=1
  EXTRA="--global-batch $gb" run b${gb}_graph GLR_GRAPH_IMG=1
done
EXTRA="--global-batch 32" run b32_forced_dist GLR_FORCE_DIST=1 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 MASTER_ADDR=127.0.0.1 MASTER_PORT=29571
EXTRA="--global-batch 32" run b32_forced_dist_graph GLR_GRAPH_IMG=1 GLR_FORCE_DIST=1 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 MASTER_ADDR=127.0.0.1 MASTER_PORT=29572
EXTRA="--global-batch 32" run b32_forced_dist_1stream GLR_ENCODER_STREAMS=0 GLR_FORCE_DIST=1 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 MASTER_ADDR=127.0.0.1 MASTER_PORT=29573
tail -5 $OUT/bench.err
