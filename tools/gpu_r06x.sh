#!/bin/bash
# round 6, call x: request delays in the TRAINING recurrences -- forward with saved state (compile-time, x 64 clocks) and backward
# (ONSSEN_BWD_DELAY of the debug-knobs build, s_sleep(1) repetitions) -- on the cfg4 training step (3 x BLSTM-600, 16 chunks)
export TMPDIR=/tmp
mkdir -p gpurun_out
out=gpurun_out/r06x_train_delay_ab.txt
: > $out
one() {  # variant, extra env
  r=$(env $2 ONSSEN_HIP_LIB=$PWD/build_variants/libonssen_hip_$1.so timeout 300 python tools/train_step_bench.py --layers 3 --steps 30 --warmup 5 2>/dev/null | tail -1 | python -c "import json,sys; r=json.loads(sys.stdin.read()); print('train step ms %.4f' % r['ms_per_step'])")
  echo "$1 $2 $r" | tee -a $out
}
for rep in 1 2; do
  one knobs ONSSEN_BWD_DELAY=0
  one t2 ONSSEN_BWD_DELAY=0
  one t4 ONSSEN_BWD_DELAY=0
  one t8 ONSSEN_BWD_DELAY=0
  one knobs ONSSEN_BWD_DELAY=2
  one knobs ONSSEN_BWD_DELAY=4
  one knobs ONSSEN_BWD_DELAY=8
  one knobs ONSSEN_BWD_DELAY=12
done
