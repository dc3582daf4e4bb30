#!/bin/bash
export TMPDIR=/tmp
timeout 200 python tools/gemm_probe.py
echo "--- x3p"; ONSSEN_X3Q=0 timeout 200 python tools/gemm_probe.py
