#!/bin/bash
export TMPDIR=/tmp
timeout 120 python tools/xcd_timeline.py 2>&1 | grep -v amdgpu.ids | tail -2
B=64 timeout 120 python tools/xcd_timeline.py 2>&1 | grep -v amdgpu.ids | tail -2
timeout 600 python -m pytest tests -m gpu -x -q 2>&1 | tail -2
for c in dc_l2 chimera_l4; do
timeout 300 python bench.py --config $c --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/bench_$c.json 2> gpurun_out/bench_$c.err
python - <<PY
import json
r = json.loads(open("gpurun_out/bench_$c.json").read().strip().splitlines()[-1])
print("$c", "ms/step", r["ms_per_step"], "xRT", r["value"], "rec us/step", r["roofline"].get("us_per_time_step"))
PY
done
