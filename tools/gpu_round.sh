#!/bin/bash
export TMPDIR=/tmp
for rg in 16 8 4; do ONSSEN_XCD_RG=$rg timeout 120 python tools/xcd_timeline.py 2>&1 | grep -v amdgpu.ids | tail -2; done
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1; tail -3 gpurun_out/pytest_gpu.log
for c in dc_l2 dc_l3 chimera_l4; do
timeout 300 python bench.py --config $c --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/bench_$c.json 2> gpurun_out/bench_$c.err
python - <<PY
import json
r = json.loads(open("gpurun_out/bench_$c.json").read().strip().splitlines()[-1])
print("$c", "ms/step", r["ms_per_step"], "xRT", r["value"], "rec us/step", r["roofline"].get("us_per_time_step"), "safe", r["config"].get("xcd_placement_independent_protocol_used"))
PY
done
