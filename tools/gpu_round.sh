#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
AB=0 timeout 300 python tools/xcd_timeline.py 2>&1 | grep -v amdgpu.ids | tail -1
echo "== pytest gpu" ; timeout 900 python -m pytest tests -m gpu -q --no-header -rf > gpurun_out/pytest_gpu.log 2>&1 ; grep "^FAILED\|Error" gpurun_out/pytest_gpu.log | cut -c1-200 | head -8; tail -1 gpurun_out/pytest_gpu.log
for x in 1 0; do
ONSSEN_XCD=$x timeout 300 python bench.py --no-cpu-baseline 2>&1 | tail -1 > gpurun_out/bench_xcd$x.log
python - <<PY
import json
r = json.loads(open("gpurun_out/bench_xcd$x.log").read().strip().splitlines()[-1]); ro = r["roofline"]
print("XCD=$x", "ms/step", round(r["ms_per_step"], 3), "xRT", round(r["value"]), "rec us/step", round(ro["us_per_time_step"], 2))
PY
done
