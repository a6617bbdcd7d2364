#!/bin/bash
timeout -k 10 300 python tools/ablate_k1_t1.py 2>&1 | grep -v amdgpu.ids | grep "full\|everything\|empty"
