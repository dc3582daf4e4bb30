#!/bin/bash
export TMPDIR=/tmp
timeout 600 python -m pytest tests -m gpu -q -x -k "cluster or ragged or separation or separate or robust" > gpurun_out/pytest_gpu_quick.log 2>&1; tail -3 gpurun_out/pytest_gpu_quick.log
timeout 500 python bench.py --no-extra --no-cpu-baseline > gpurun_out/bench_quick.json 2> gpurun_out/bench_quick.err; tail -c 300 gpurun_out/bench_quick.err
python - <<'PY'
import json
r = json.loads(open("gpurun_out/bench_quick.json").read().strip().splitlines()[-1])
print("headline ms/step", r["ms_per_step"], "resident", r["resident_mask_step"]["ms_per_step"])
print(r["roofline"].get("dc_back_end_legs_ms"))
print(r["roofline"]["legs_ms"])
PY
