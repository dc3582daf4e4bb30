#!/bin/bash
# Scratch wrapper for one gpurun call while iterating (edit freely).
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_robustness.py tests/test_gpu_parity.py -m gpu -x -q -k "robust or abort or cotenant or rccl or nonfinite or dc_cluster or chimera_loss or e2e or separ or recipe" > gpurun_out/pytest_gpu.log 2>&1; tail -8 gpurun_out/pytest_gpu.log
timeout 400 python bench.py > gpurun_out/bench_r03b.json 2> gpurun_out/bench_r03b.err; python - <<'PY'
import json
r=json.loads(open('gpurun_out/bench_r03b.json').read().strip().splitlines()[-1])
ro=r["roofline"]
print("headline", r["ms_per_step"], "rec us/step", ro["us_per_time_step"], "legs", ro.get("legs_ms"), ro.get("legs_sum_ms"), ro.get("legs_le_step"))
print("km", r.get("separate_dc_with_device_kmeans"))
PY
tail -3 gpurun_out/bench_r03b.err
HEAVY=1 timeout 300 python tools/cotenant_probe.py > gpurun_out/cotenant_probe_heavy.txt 2> gpurun_out/cotenant_probe.err; cat gpurun_out/cotenant_probe_heavy.txt; tail -5 gpurun_out/cotenant_probe.err
