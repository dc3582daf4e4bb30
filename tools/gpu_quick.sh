#!/bin/bash
export TMPDIR=/tmp
q() { python -c "import json,sys; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', round(r['ms_per_step'],4), round(r['roofline']['us_per_time_step'],3))"; }
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q 2>&1 | tail -4
for dc in 0; do for di in 0 3; do
ONSSEN_XCD_DELAY_CELL=$dc ONSSEN_XCD_DELAY_IDLE=$di timeout 100 python bench.py --no-cpu-baseline 2>&1 | q "dc_l2 delay cell=$dc idle=$di"
done; done
timeout 100 python bench.py --no-cpu-baseline --config chimera_l4 2>&1 | q "chimera"
timeout 100 python bench.py --no-cpu-baseline --config dc_l3 2>&1 | q "dc_l3"
timeout 100 python bench.py --no-cpu-baseline --config phase_l4 2>&1 | q "phase_l4"
