#!/bin/bash
# round 6, call k: same-box comparison of the round-5 tree (build_variants/r05_tree, commit a4f197d) with HEAD: headline, chimera B=64, training step
export TMPDIR=/tmp
mkdir -p gpurun_out
one() { # tree, config
  (cd $1 && timeout 600 python bench.py --no-cpu-baseline --no-extra --steps 40 --config $2 2>/dev/null | python -c "
import json,sys; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1 $2 ms_per_step %.4f' % r['ms_per_step'])")
}
for rep in 1 2; do
  for cfg in dc_l2 chimera_l4; do
    one build_variants/r05_tree $cfg
    one . $cfg
  done
  (cd build_variants/r05_tree && timeout 300 python tools/train_step_bench.py --layers 3 --steps 30 --warmup 5 2>/dev/null | tail -1 | python -c "
import json,sys; r=json.loads(sys.stdin.read()); print('r05 train dc_l3 ms/step %.3f' % r['ms_per_step'])")
  timeout 300 python tools/train_step_bench.py --layers 3 --steps 30 --warmup 5 2>/dev/null | tail -1 | python -c "
import json,sys; r=json.loads(sys.stdin.read()); print('HEAD train dc_l3 ms/step %.3f' % r['ms_per_step'])"
done
