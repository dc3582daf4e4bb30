#!/bin/bash
# round 6c, call 1: the ragged pipeline's GPU tests, the whole pipeline test file, the ABI test
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_pipeline.py tests/test_gpu_ragged.py -m gpu -q -x > gpurun_out/r06c1_pytest.log 2>&1; echo "pytest rc $?"; tail -25 gpurun_out/r06c1_pytest.log
