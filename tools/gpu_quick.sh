#!/bin/bash
export TMPDIR=/tmp
for mode in 1 0; do
  ONSSEN_TRAIN_HIP=$mode timeout 200 python tools/train_step_bench.py --layers 3 --steps 10 --warmup 3 --dropout 0 2>&1 < /dev/null | tail -1 > gpurun_out/train_nodrop_hip$mode.json
  timeout 20 python -c "
import json; r=json.load(open('gpurun_out/train_nodrop_hip$mode.json')); print('train nodrop hip=$mode', r['ms_per_step'], r['value'], r['last_loss'])"
done
