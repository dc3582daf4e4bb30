#!/bin/bash
export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "training" 2>&1 | grep -E "^E  |passed|failed|FAILED" | cut -c1-600 | head -12
for mode in x3; do
  ONSSEN_TRAIN_GEMM=$mode timeout 200 python tools/train_step_bench.py --layers 3 --steps 10 --warmup 3 2>&1 | tail -1 > gpurun_out/train_gemm_$mode.json
  timeout 20 python -c "
import json; r=json.load(open('gpurun_out/train_gemm_$mode.json')); print('train gemm=$mode', r['ms_per_step'], r['value'], r['last_loss'])"
done
