#!/bin/bash
# round 6, call e: stamped timelines (base / input projection early / no gate math), ablated variants with the early projection, and
# the 8-rank self-launch record of bench.py on one device (harness self-test)
export TMPDIR=/tmp
mkdir -p gpurun_out
for v in pbase pg1 pabl; do
  echo "== $v"; STN=32 ONSSEN_HIP_LIB=$PWD/build_variants/libonssen_hip_$v.so timeout 300 python tools/xcd_timeline.py 2>&1 | tail -4
done > gpurun_out/r06e_timeline.txt 2>&1
cat gpurun_out/r06e_timeline.txt
python tools/ab_variants.py run base g1 g1d2a g1d4a g1d6a abl192 -- bench.py --no-cpu-baseline --no-extra --steps 40 > gpurun_out/r06e_ab.txt 2>&1; cat gpurun_out/r06e_ab.txt
ONSSEN_BENCH_ONE_DEVICE=1 timeout 1200 python bench.py --gpus 8 --steps 3 --warmup 1 > gpurun_out/r06_bench_selflaunch_8rank.json 2> gpurun_out/r06_bench_selflaunch_8rank.err; echo "8-rank rc $?"; tail -c 1500 gpurun_out/r06_bench_selflaunch_8rank.json; tail -5 gpurun_out/r06_bench_selflaunch_8rank.err
