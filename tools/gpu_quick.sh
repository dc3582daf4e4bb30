#!/bin/bash
# Scratch wrapper for one gpurun call while iterating (edit freely): a few parity tests + the headline bench.
export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "golden or xcd or cfg" 2>&1 | tail -2
timeout 100 python bench.py --no-cpu-baseline --steps 40 | python -c "import sys,json; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(r['ms_per_step'], r['roofline']['us_per_time_step'])"
