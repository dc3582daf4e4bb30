#!/bin/bash
# One gpurun round trip: smoke, GPU parity tests, bench, rocprof kernel stats.  Outputs under gpurun_out/.
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== smoke" ; timeout 300 python -c "import __graft_entry__ as g; g.build(); g.smoke()" 2>&1 | tail -5
echo "== pytest gpu" ; timeout 900 python -m pytest tests -m gpu -q --no-header -rf 2>&1 | tail -40 | tee gpurun_out/pytest_gpu.log | tail -25
echo "== bench" ; timeout 600 python bench.py --steps 10 --warmup 3 2>&1 | tail -3 | tee gpurun_out/bench.log
