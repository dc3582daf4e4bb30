#!/bin/bash
# Scratch wrapper for one gpurun call while iterating (edit freely).
export TMPDIR=/tmp
for v in "ONSSEN_X3Q_BM=128" "ONSSEN_X3Q=256" "ONSSEN_X3Q=0"; do echo "== $v"; env $v timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "golden or cfg or linear or gemm or e2e or separ" 2>&1 | tail -2; done
