#!/bin/bash
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1; tail -4 gpurun_out/pytest_gpu.log
timeout 400 python bench.py > gpurun_out/bench_r03d.json 2> gpurun_out/bench_r03d.err; python - <<'PY'
import json
r=json.loads(open('gpurun_out/bench_r03d.json').read().strip().splitlines()[-1])
ro=r["roofline"]
print("headline", r["ms_per_step"], "xRT", r["value"], "rec us/step", ro["us_per_time_step"], "frac", ro["frac"])
print("legs", ro.get("legs_ms"), ro.get("legs_sum_ms"), ro.get("legs_le_step")); print(ro.get("first_layer"))
print("km", r.get("separate_dc_with_device_kmeans", {}).get("ms_per_step"))
for k,v in r["extra_configs"].items(): print(k, {a:b for a,b in v.items() if a in ("ms_per_step","error")})
PY
tail -3 gpurun_out/bench_r03d.err
