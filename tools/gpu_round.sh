#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== pytest gpu" ; timeout 1200 python -m pytest tests -m gpu -q --no-header -rf > gpurun_out/pytest_gpu.log 2>&1 ; grep "^FAILED" gpurun_out/pytest_gpu.log | cut -c1-150; tail -2 gpurun_out/pytest_gpu.log
echo "== timeline" ; timeout 600 python tools/step_timeline.py 2>&1 | grep -v amdgpu.ids | tail -7 | tee gpurun_out/timeline.log
echo "== bench default" ; timeout 900 python bench.py --no-cpu-baseline 2>&1 | tail -1 | tee gpurun_out/bench.log | cut -c1-250
