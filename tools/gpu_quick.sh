#!/bin/bash
export TMPDIR=/tmp
ONSSEN_X3P_WDIRECT=1 timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "x3_image" 2>&1 | tail -2
timeout 200 python tools/gemm_probe2.py 2>&1 | grep -v amdgpu.ids | tail -5
ONSSEN_X3P_WDIRECT=1 timeout 200 python tools/gemm_probe2.py 2>&1 | grep -v amdgpu.ids | tail -5
