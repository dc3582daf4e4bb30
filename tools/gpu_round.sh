#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== pytest gpu" ; timeout 1200 python -m pytest tests -m gpu -q --no-header -rf -s > gpurun_out/pytest_gpu.log 2>&1 ; grep "split-bf16" gpurun_out/pytest_gpu.log | sort | uniq; tail -3 gpurun_out/pytest_gpu.log
echo "== bench f32" ; timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 | tee gpurun_out/bench_f32.log | cut -c1-330
echo "== bench bf16x3" ; timeout 600 python bench.py --precision bf16x3 --steps 10 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 | tee gpurun_out/bench_x3.log | cut -c1-330
python - <<'PY'
import json
for f in ("gpurun_out/bench_f32.log", "gpurun_out/bench_x3.log"):
    try:
        r = json.loads(open(f).read().strip().splitlines()[-1])["roofline"]
        print(f, "rec us/step", round(r["us_per_time_step"], 2), "frac", round(r["frac"], 4), r["other_kernels"]["achieved_by_call"], r["other_kernels"]["ms_by_call"])
    except Exception as e:
        print(f, "ERR", e)
PY
