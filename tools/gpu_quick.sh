#!/bin/bash
export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "training" 2>&1 | grep -v "^$" | tail -5
for mode in 1; do
  ONSSEN_TRAIN_HIP=$mode timeout 300 python tools/train_step_bench.py --layers 3 --steps 10 --warmup 3 2>&1 | tail -1 > gpurun_out/train_hip$mode.json
  timeout 20 python -c "
import json; r=json.load(open('gpurun_out/train_hip$mode.json')); print('train hip=$mode', r['ms_per_step'], r['value'], r['last_loss'])"
done
