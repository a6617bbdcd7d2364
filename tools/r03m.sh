#!/bin/bash
run() { name=$1; shift; env "$@" timeout -k 10 400 python bench.py --no-cpu-baseline --steps 10 2>>gpurun_out/r03m.err | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$name:', round(d['ms_per_step'],2), 'ms/step first', round(d['config']['first_step_s'],1))"; }
run base A=1
run no_asm_bwd MIOPEN_DEBUG_CONV_IMPLICIT_GEMM_ASM_BWD_GTC_DYNAMIC_XDLOPS_NHWC=0
run no_asm_wrw MIOPEN_DEBUG_CONV_IMPLICIT_GEMM_ASM_WRW_GTC_DYNAMIC_XDLOPS_NHWC=0
run no_asm_bwd_wrw MIOPEN_DEBUG_CONV_IMPLICIT_GEMM_ASM_BWD_GTC_DYNAMIC_XDLOPS_NHWC=0 MIOPEN_DEBUG_CONV_IMPLICIT_GEMM_ASM_WRW_GTC_DYNAMIC_XDLOPS_NHWC=0
run no_ck_bwd_wrw MIOPEN_DEBUG_CONV_IMPLICIT_GEMM_HIP_GROUP_BWD_XDLOPS=0 MIOPEN_DEBUG_CONV_IMPLICIT_GEMM_HIP_GROUP_WRW_XDLOPS=0
