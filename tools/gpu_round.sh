#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
for gn in 1 2 4 6 30; do ONSSEN_X3_GN=$gn timeout 120 python tools/gemm_probe.py 2>&1 | tail -1 | sed "s/^/GN=$gn /"; done
for gn in 30 4; do
ONSSEN_X3_GN=$gn timeout 900 python bench.py --no-cpu-baseline 2>&1 | tail -1 > gpurun_out/bench_gn$gn.log
python - <<PY
import json
r = json.loads(open("gpurun_out/bench_gn$gn.log").read().strip().splitlines()[-1]); ro = r["roofline"]
print("GN=$gn", "ms/step", round(r["ms_per_step"], 3), "xRT", round(r["value"]), ro["other_kernels"]["ms_by_call"])
PY
done
echo "== pytest gpu" ; timeout 1200 python -m pytest tests -m gpu -q --no-header -rf > gpurun_out/pytest_gpu.log 2>&1 ; grep "^FAILED" gpurun_out/pytest_gpu.log | cut -c1-150; tail -2 gpurun_out/pytest_gpu.log
