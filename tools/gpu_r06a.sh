#!/bin/bash
# round 6, call a: the new feature_utils names + the whole GPU suite + the default bench line (baseline of the round)
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 300 python -m pytest tests/test_gpu_feature_utils.py -m gpu -q -x -s > gpurun_out/r06a_feature_utils.log 2>&1; echo "feature_utils rc $?"; tail -4 gpurun_out/r06a_feature_utils.log
timeout 1500 python -m pytest tests -m gpu -q -x > gpurun_out/r06a_pytest_gpu.log 2>&1; echo "pytest rc $?"; tail -3 gpurun_out/r06a_pytest_gpu.log
timeout 900 python bench.py > gpurun_out/r06a_bench.json 2> gpurun_out/r06a_bench.err; echo "bench rc $?"; python - <<'P'
import json
r = json.loads(open('gpurun_out/r06a_bench.json').read().strip().splitlines()[-1])
print('ms_per_step', r['ms_per_step'], 'value', r['value'], 'roofline', {k: r['roofline'].get(k) for k in ('achieved', 'frac', 'us_per_time_step')})
print('cpu_baseline', r.get('cpu_baseline'))
P
