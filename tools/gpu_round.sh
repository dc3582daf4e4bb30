#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
echo "== pytest gpu" ; timeout 1200 python -m pytest tests -m gpu -q --no-header -rf > gpurun_out/pytest_gpu.log 2>&1 ; grep "^FAILED\|Error" gpurun_out/pytest_gpu.log | cut -c1-200 | head; tail -2 gpurun_out/pytest_gpu.log
