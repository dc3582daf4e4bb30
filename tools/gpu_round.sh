#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== pytest gpu" ; timeout 1200 python -m pytest tests -m gpu -q --no-header -rf > gpurun_out/pytest_gpu.log 2>&1 ; grep "^FAILED" gpurun_out/pytest_gpu.log | cut -c1-150; tail -2 gpurun_out/pytest_gpu.log
for prec in f32 bf16x3; do
timeout 900 python bench.py --precision $prec --no-cpu-baseline 2>&1 | tail -1 > gpurun_out/bench_$prec.log
python - <<PY
import json
r = json.loads(open("gpurun_out/bench_$prec.log").read().strip().splitlines()[-1]); ro = r["roofline"]
print("$prec", "ms/step", round(r["ms_per_step"], 3), "xRT", round(r["value"]), "rec us/step", round(ro["us_per_time_step"], 2), "frac", round(ro["frac"], 4))
PY
done
