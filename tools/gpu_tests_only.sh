#!/bin/bash
export TMPDIR=/tmp
timeout 1400 python -m pytest tests -m gpu -q "$@" > gpurun_out/pytest_gpu.log 2>&1; tail -15 gpurun_out/pytest_gpu.log
grep -E "^\[(bf16|cfg4)" -A12 gpurun_out/pytest_gpu.log | head -70
