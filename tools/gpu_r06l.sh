#!/bin/bash
# round 6, call l: the backward recurrence's wide poll (16-byte loads, 16 lanes per unit) against the round-2 poll, same box
export TMPDIR=/tmp
mkdir -p gpurun_out
for rep in 1 2; do for v in base wide; do
  ONSSEN_HIP_LIB=$PWD/build_variants/libonssen_hip_$v.so timeout 300 python tools/train_step_bench.py --layers 3 --steps 30 --warmup 5 2>/dev/null | tail -1 | python -c "
import json,sys; r=json.loads(sys.stdin.read()); print('$v train dc_l3 b16 ms/step %.3f loss %.4f' % (r['ms_per_step'], r['last_loss']))"
done; done
for v in base wide; do
  ONSSEN_HIP_LIB=$PWD/build_variants/libonssen_hip_$v.so timeout 600 python bench.py --no-cpu-baseline --steps 10 2>/dev/null | python -c "
import json,sys; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); t=r['extra_configs']['cfg4_training_step_dc_l3_b16']; print('$v', 'cfg4 ms', t['ms_per_step'], {k: t[k].get('us_per_time_step') for k in t if isinstance(t[k], dict) and 'us_per_time_step' in t[k]})"
done
ONSSEN_HIP_LIB=$PWD/build_variants/libonssen_hip_wide.so timeout 1500 python -m pytest tests -m gpu -q -x -k "train or grad or backward or cfg4 or loss or recipe or adam or wide or robust" 2>&1 | tail -4
