#!/bin/bash
# fresh MIOpen find (perf-db only, no find-db) with solver families excluded: does the step get faster without the
# asm GTC solvers' zero fills?
mkdir -p gpurun_out
run() { name=$1; shift; d=/tmp/mdb_$name; rm -rf $d; mkdir -p $d; cp gloria-nlp-project_amd/miopen_db/*.udb.txt $d/
  env MIOPEN_USER_DB_PATH=$d "$@" timeout -k 10 500 python bench.py --no-cpu-baseline --steps 10 2>>gpurun_out/r03n.err | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$name:', round(d['ms_per_step'],2), 'ms/step first', round(d['config']['first_step_s'],1))"; }
run fresh A=1 &&
run no_asm_wrw MIOPEN_DEBUG_CONV_IMPLICIT_GEMM_ASM_WRW_GTC_DYNAMIC_XDLOPS_NHWC=0 &&
run no_asm_bwd MIOPEN_DEBUG_CONV_IMPLICIT_GEMM_ASM_BWD_GTC_DYNAMIC_XDLOPS_NHWC=0 &&
run no_asm_bwd_wrw MIOPEN_DEBUG_CONV_IMPLICIT_GEMM_ASM_BWD_GTC_DYNAMIC_XDLOPS_NHWC=0 MIOPEN_DEBUG_CONV_IMPLICIT_GEMM_ASM_WRW_GTC_DYNAMIC_XDLOPS_NHWC=0
