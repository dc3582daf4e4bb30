#!/bin/bash
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1; tail -3 gpurun_out/pytest_gpu.log
timeout 200 python tools/xcd_graph_probe.py 2>&1 | tail -3
for c in dc_l2 dc_l3 chimera_l4; do
for f in 1 0; do
ONSSEN_FUSE_IN0=$f timeout 300 python bench.py --config $c --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/bench_$c.json 2> gpurun_out/bench_$c.err
python - <<PY
import json
r = json.loads(open("gpurun_out/bench_$c.json").read().strip().splitlines()[-1])
print("$c fuse=$f", "ms/step", r["ms_per_step"], "xRT", r["value"], "rec us/step", r["roofline"].get("us_per_time_step"), "safe", r["config"].get("xcd_placement_independent_protocol_used"))
PY
done
done
