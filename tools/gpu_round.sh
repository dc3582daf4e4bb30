#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== pytest gpu" ; timeout 1200 python -m pytest tests -m gpu -q --no-header -rf > gpurun_out/pytest_gpu.log 2>&1 ; grep "^FAILED" gpurun_out/pytest_gpu.log | cut -c1-150; tail -2 gpurun_out/pytest_gpu.log
echo "== bench default" ; timeout 900 python bench.py --no-cpu-baseline 2>&1 | tail -1 | tee gpurun_out/bench.log | cut -c1-200
python - <<'PY'
import json
r = json.loads(open("gpurun_out/bench.log").read().strip().splitlines()[-1]); ro = r["roofline"]
print("ms/step", round(r["ms_per_step"], 3), "rec us/step", round(ro["us_per_time_step"], 2), ro["other_kernels"]["achieved_by_call"], ro["other_kernels"]["ms_by_call"])
PY
