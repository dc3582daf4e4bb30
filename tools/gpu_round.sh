#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== pytest gpu" ; timeout 1200 python -m pytest tests -m gpu -q --no-header -rf > gpurun_out/pytest_gpu.log 2>&1 ; tail -3 gpurun_out/pytest_gpu.log
echo "== timeline" ; timeout 600 python tools/step_timeline.py 2>&1 | grep -v amdgpu.ids | tail -4 | tee gpurun_out/timeline.log
echo "== ablate" ; timeout 600 python tools/ablate_step.py 2>&1 | grep -v amdgpu.ids | sed -n 1,2p | tee gpurun_out/ablate.log
echo "== bench f32" ; timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-330
echo "== bench bf16x3" ; ONSSEN_PRECISION=bf16x3 timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-330
