#!/bin/bash
export TMPDIR=/tmp
timeout 400 python tools/train_soak.py --steps 400 2>&1 | tail -1 | cut -c1-300
timeout 300 python tools/xcd_soak.py 2>&1 | tail -2 | cut -c1-300
for i in 1 2 3; do timeout 100 python bench.py --no-cpu-baseline --steps 200 | python -c "import sys,json; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(r['ms_per_step'], r['roofline']['us_per_time_step'], {k:v for k,v in r['config'].items() if 'status' in k or 'abort' in k or 'protocol' in k})"; done
