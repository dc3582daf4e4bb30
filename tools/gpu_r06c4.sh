#!/bin/bash
# round 6c, call 4: the final tree once more -- smoke, whole GPU suite, default bench line
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r06c4_smoke.log 2>&1; echo "smoke rc $?"; tail -1 gpurun_out/r06c4_smoke.log
timeout 2400 python -m pytest tests -m gpu -q -x > gpurun_out/r06c4_pytest_gpu.log 2>&1; echo "pytest rc $?"; tail -2 gpurun_out/r06c4_pytest_gpu.log
timeout 1200 python bench.py > gpurun_out/r06c4_bench.json 2> gpurun_out/r06c4_bench.err; echo "bench rc $?"; python - <<'P'
import json
r = json.loads(open('gpurun_out/r06c4_bench.json').read().strip().splitlines()[-1])
print('steps', r['steps'], 'warmup', r['warmup'], 'ms_per_step', r['ms_per_step'], 'value', r['value'], 'frac', r['roofline']['frac'], 'traffic', r['roofline']['traffic'])
P
