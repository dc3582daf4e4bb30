#!/bin/bash
# Scratch entry point for short gpurun calls during development (edit freely); the round-end validation is tools/gpu_round.sh.
export TMPDIR=/tmp
ONSSEN_XCD_WAVES=8 timeout 100 python bench.py --no-cpu-baseline 2>&1 | tail -1 | python -c "import json,sys; r=json.loads(sys.stdin.read()); print('8 waves dc_l2', r['ms_per_step'], r['roofline']['us_per_time_step'])"
ONSSEN_XCD_WAVES=8 timeout 100 python bench.py --no-cpu-baseline --config chimera_l4 2>&1 | tail -1 | python -c "import json,sys; r=json.loads(sys.stdin.read()); print('8 waves chimera', r['ms_per_step'], r['roofline']['us_per_time_step'])"
timeout 100 python bench.py --no-cpu-baseline 2>&1 | tail -1 | python -c "import json,sys; r=json.loads(sys.stdin.read()); print('default dc_l2', r['ms_per_step'], r['roofline']['us_per_time_step'])"
timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "xcd or golden or graph or training" 2>&1 | tail -2
