#!/bin/bash
export TMPDIR=/tmp
for ab in clk0 clk11 clk3 clk1 clk4; do ONSSEN_PROBE_LIB=build_variants/lib_$ab.so timeout 120 python tools/gemm_probe3.py 2>&1 | grep -v amdgpu.ids | tail -2; done
