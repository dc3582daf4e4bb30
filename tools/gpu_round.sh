#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
echo "== pytest gpu" ; timeout 1200 python -m pytest tests -m gpu -q --no-header -rf > gpurun_out/pytest_gpu.log 2>&1 ; grep "^FAILED" gpurun_out/pytest_gpu.log | cut -c1-150; tail -2 gpurun_out/pytest_gpu.log
for wm in 2 4; do for ab in 0 15; do ONSSEN_X3_WM=$wm ONSSEN_X3_ABLATE=$ab timeout 120 python tools/gemm_probe.py 2>&1 | tail -1; done; done
for wm in 2 4; do
ONSSEN_X3_WM=$wm timeout 900 python bench.py --no-cpu-baseline 2>&1 | tail -1 > gpurun_out/bench_wm$wm.log
python - <<PY
import json
r = json.loads(open("gpurun_out/bench_wm$wm.log").read().strip().splitlines()[-1]); ro = r["roofline"]
print("WM=$wm", "ms/step", round(r["ms_per_step"], 3), "xRT", round(r["value"]), ro["other_kernels"]["ms_by_call"])
PY
done
