#!/bin/bash
# Scratch entry point for short gpurun calls during development (edit freely); the round-end validation is tools/gpu_round.sh.
export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "training" 2>&1 | tail -3
