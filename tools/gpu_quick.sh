#!/bin/bash
export TMPDIR=/tmp
for c in dc_l2 dc_l3 phase_l4; do for f in auto 1 auto 1; do echo -n "$c fuse=$f: "; ONSSEN_FUSE_IN0=$f timeout 100 python bench.py --no-cpu-baseline --steps 40 --config $c | python -c "import sys,json; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(r['ms_per_step'], r['roofline']['us_per_time_step'])"; done; done
