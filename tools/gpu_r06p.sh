#!/bin/bash
# round 6, call p: schedule knobs of the recurrence kernel re-measured on the PAIR launch (16-row groups), same box, interleaved
export TMPDIR=/tmp
mkdir -p gpurun_out
out=gpurun_out/${OUT:-r06p_pair_ab.txt}
: > $out
for rep in 1 2; do
  for v in ${VARIANTS:-base skipoor gearly fetchfirst tg221 reqearly}; do
    f=build_variants/libonssen_hip_$v.so
    [ -f $f ] || continue
    r=$(ONSSEN_HIP_LIB=$PWD/$f timeout 200 python tools/micro/pipe2_profile.py 60 2>&1 | tail -1)
    echo "$v $r" | tee -a $out
  done
done
