#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
echo "== pytest xcd" ; timeout 600 python -m pytest tests -m gpu -q --no-header -rf -s -k "xcd_local" > gpurun_out/pytest_xcd.log 2>&1 ; grep "xcd-local\|^FAILED\|Error\|abort" gpurun_out/pytest_xcd.log | cut -c1-220 | head -12; tail -2 gpurun_out/pytest_xcd.log
for x in 1 0; do
ONSSEN_XCD=$x timeout 300 python bench.py --no-cpu-baseline 2>&1 | tail -1 > gpurun_out/bench_xcd$x.log
python - <<PY
import json
try:
    r = json.loads(open("gpurun_out/bench_xcd$x.log").read().strip().splitlines()[-1]); ro = r["roofline"]
    print("XCD=$x", "ms/step", round(r["ms_per_step"], 3), "xRT", round(r["value"]), "rec us/step", round(ro["us_per_time_step"], 2))
except Exception as e:
    print("XCD=$x ERR", e, open("gpurun_out/bench_xcd$x.log").read()[-400:])
PY
done
