#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
for cfg in dc_l2 dc_l3 chimera_l4; do
echo "== bench $cfg" ; timeout 900 python bench.py --config $cfg --no-cpu-baseline 2>&1 | tail -1 > gpurun_out/bench_$cfg.log
python - <<PY
import json
r = json.loads(open("gpurun_out/bench_$cfg.log").read().strip().splitlines()[-1]); ro = r["roofline"]
print("$cfg", "ms/step", round(r["ms_per_step"], 3), "xRT", round(r["value"]), "frames/s", round(r["frames_per_s"]), "rec us/step", round(ro["us_per_time_step"], 2), "frac", round(ro["frac"],4))
PY
done
echo "== bench f32"; timeout 900 python bench.py --precision f32 --no-cpu-baseline 2>&1 | tail -1 > gpurun_out/bench_f32.log
python - <<PY
import json
r = json.loads(open("gpurun_out/bench_f32.log").read().strip().splitlines()[-1]); ro = r["roofline"]
print("f32 dc_l2", "ms/step", round(r["ms_per_step"], 3), "xRT", round(r["value"]), "rec us/step", round(ro["us_per_time_step"], 2), "frac", round(ro["frac"],4))
PY
