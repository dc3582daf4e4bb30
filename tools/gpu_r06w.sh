#!/bin/bash
# round 6, call w: (1) fetch-first / early input projection on top of the new unstacked defaults (pipelined step);
# (2) the request delay of the cell-update waves on the STACKED 8-row groups (one-batch step, recurrence us per time step)
export TMPDIR=/tmp
mkdir -p gpurun_out
out=gpurun_out/r06w_ab.txt
: > $out
for rep in 1 2; do
  for v in base uff uge; do
    r=$(ONSSEN_HIP_LIB=$PWD/build_variants/libonssen_hip_$v.so timeout 200 python tools/micro/pipe2_profile.py 60 2>&1 | tail -1)
    echo "$v $r" | tee -a $out
  done
done
for rep in 1 2; do
  for v in base dall2 dall4 dall8 dall12; do
    r=$(ONSSEN_HIP_LIB=$PWD/build_variants/libonssen_hip_$v.so timeout 300 python bench.py --no-extra --no-cpu-baseline --no-pipeline --steps 40 2>/dev/null | tail -1 | python -c "import json,sys; r=json.loads(sys.stdin.read()); print('one-batch step ms %.4f' % r['ms_per_step'], 'recurrence us/step %.4f' % r['roofline']['us_per_time_step'], 'fused l0 ms %.4f' % r['roofline']['first_layer']['ms'])")
    echo "$v $r" | tee -a $out
  done
done
