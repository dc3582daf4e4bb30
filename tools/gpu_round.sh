#!/bin/bash
# One gpurun round trip: smoke, GPU parity tests, bench (both recurrence forms), rocprof kernel stats.
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== smoke" ; timeout 300 python -c "import __graft_entry__ as g; g.build(); g.smoke()" 2>&1 | tail -3
echo "== pytest gpu" ; timeout 1200 python -m pytest tests -m gpu -q --no-header -rf -x > gpurun_out/pytest_gpu.log 2>&1 ; tail -12 gpurun_out/pytest_gpu.log
echo "== bench persistent" ; timeout 600 python bench.py --steps 10 --warmup 3 2>&1 | tail -1 | tee gpurun_out/bench.log
echo "== bench launch-per-step" ; ONSSEN_PERSISTENT=0 timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 | tee gpurun_out/bench_v1.log
echo "== rocprof" ; cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/rocprof.log 2>&1; cd $GRAFT_REPO_ROOT
f=$(find gpurun_out/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cut -c1-200 "$f" | head -12
find gpurun_out/prof -name "*kernel_trace*" -size +1M -delete
