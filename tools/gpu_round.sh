#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 300 python tools/xcd_timeline.py 2>&1 | grep -v amdgpu.ids | tail -2 | tee gpurun_out/xcd_timeline.log
echo "== pytest xcd" ; timeout 600 python -m pytest tests -m gpu -q --no-header -rf -s -k "xcd_local" > gpurun_out/pytest_xcd.log 2>&1 ; grep "xcd-local\|^FAILED\|Error\|abort" gpurun_out/pytest_xcd.log | cut -c1-220 | head -8; tail -1 gpurun_out/pytest_xcd.log
ONSSEN_XCD=1 timeout 300 python bench.py --no-cpu-baseline 2>&1 | tail -1 > gpurun_out/bench_xcd1.log
python - <<PY
import json
r = json.loads(open("gpurun_out/bench_xcd1.log").read().strip().splitlines()[-1]); ro = r["roofline"]
print("XCD=1", "ms/step", round(r["ms_per_step"], 3), "xRT", round(r["value"]), "rec us/step", round(ro["us_per_time_step"], 2))
PY
