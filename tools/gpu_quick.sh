#!/bin/bash
# Scratch wrapper for one gpurun call while iterating (edit freely).
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1; tail -6 gpurun_out/pytest_gpu.log
for p in f32; do for x in 1 0; do echo "precision=$p XCD=$x"; ONSSEN_XCD=$x timeout 200 python bench.py --precision $p --no-cpu-baseline --no-extra --steps 20 2>/dev/null | python -c "import json,sys; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(r['ms_per_step'], r['roofline']['us_per_time_step'], r['roofline']['frac'], r['config']['recurrence'], r['roofline']['legs_ms'])"; done; done
