#!/bin/bash
# One gpurun round trip: smoke, GPU parity tests, bench, rocprof kernel stats.  Outputs under gpurun_out/.
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== smoke" ; timeout 300 python -c "import __graft_entry__ as g; g.build(); g.smoke()" 2>&1 | tail -5
echo "== pytest gpu" ; timeout 900 python -m pytest tests -m gpu -q --no-header -rf > gpurun_out/pytest_gpu.log 2>&1 ; tail -8 gpurun_out/pytest_gpu.log
echo "== bench" ; timeout 600 python bench.py --steps 10 --warmup 3 2>&1 | tail -3 | tee gpurun_out/bench.log
echo "== rocprof" ; cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/rocprof.log 2>&1; cd $GRAFT_REPO_ROOT
find gpurun_out/prof -name "*kernel_stats*" | head; f=$(find gpurun_out/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -20 "$f"
find gpurun_out/prof -name "*kernel_trace*" -size +1M -delete
