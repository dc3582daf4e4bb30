#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== bench default" ; timeout 900 python bench.py 2>&1 | tail -1 | tee gpurun_out/bench.log | cut -c1-250
echo "== bench f32" ; timeout 600 python bench.py --precision f32 --no-cpu-baseline 2>&1 | tail -1 | tee gpurun_out/bench_f32.log | cut -c1-250
echo "== rocprof" ; cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/rocprof.log 2>&1; cd $GRAFT_REPO_ROOT
f=$(find gpurun_out/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cut -c1-160 "$f" | head -9
find gpurun_out/prof -name "*kernel_trace*" -size +1M -delete
