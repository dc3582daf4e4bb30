#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_trained.py -m gpu -q -x -k weight_guard 2>&1 | tail -25
timeout 900 python bench.py --no-cpu-baseline --no-extra --steps 40 2>/dev/null | python -c "
import json,sys; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('guard on  ms_per_step', r['ms_per_step'])"
ONSSEN_WEIGHT_GUARD=0 timeout 900 python bench.py --no-cpu-baseline --no-extra --steps 40 2>/dev/null | python -c "
import json,sys; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('guard off ms_per_step', r['ms_per_step'])"
timeout 900 python bench.py --no-cpu-baseline --no-extra --steps 40 2>/dev/null | python -c "
import json,sys; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('guard on  ms_per_step', r['ms_per_step'])"
ONSSEN_WEIGHT_GUARD=0 timeout 900 python bench.py --no-cpu-baseline --no-extra --steps 40 2>/dev/null | python -c "
import json,sys; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('guard off ms_per_step', r['ms_per_step'])"
