#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== smoke" ; timeout 300 python -c "import __graft_entry__ as g; g.build(); g.smoke()" 2>&1 | tail -2
echo "== pytest gpu" ; timeout 1200 python -m pytest tests -m gpu -q --no-header -rf -s > gpurun_out/pytest_gpu.log 2>&1 ; grep "^FAILED" gpurun_out/pytest_gpu.log | cut -c1-150; grep "max abs err" gpurun_out/pytest_gpu.log | sort | uniq | head -12; tail -2 gpurun_out/pytest_gpu.log
