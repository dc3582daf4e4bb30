#!/bin/bash
export TMPDIR=/tmp
timeout 600 python -m pytest tests -m gpu -q -x -k "cluster or ragged or separation or separate or robust or smoke" > gpurun_out/pytest_gpu_quick.log 2>&1; tail -3 gpurun_out/pytest_gpu_quick.log
timeout 300 python -c "import __graft_entry__ as g; g.build(); g.smoke()" 2>&1 | tail -1
bash tools/gpu_ab_compact.sh 2>&1 | head -24
