#!/bin/bash
# round 6, call s: s_setprio on the cell-update waves -- unstacked 16-row instantiations only (prio1u/2u/3u) or everywhere (prio2all);
# pipelined step and, for the everywhere form, the one-batch step (8-row stacked groups)
export TMPDIR=/tmp
mkdir -p gpurun_out
out=gpurun_out/r06s_prio_ab.txt
: > $out
for rep in 1 2; do
  for v in base prio1u prio2u prio3u prio2all; do
    f=build_variants/libonssen_hip_$v.so
    [ -f $f ] || continue
    r=$(ONSSEN_HIP_LIB=$PWD/$f timeout 200 python tools/micro/pipe2_profile.py 60 2>&1 | tail -1)
    echo "$v $r" | tee -a $out
  done
done
for rep in 1 2; do
  for v in base prio2all; do
    r=$(ONSSEN_HIP_LIB=$PWD/build_variants/libonssen_hip_$v.so timeout 300 python bench.py --no-extra --no-cpu-baseline --no-pipeline --steps 40 2>/dev/null | tail -1 | python -c "import json,sys; r=json.loads(sys.stdin.read()); print('one-batch step ms', r['ms_per_step'], 'recurrence us/step', r['roofline']['us_per_time_step'])")
    echo "$v $r" | tee -a $out
  done
done
