#!/bin/bash
# round 6c, call 5: the 32x32x16-MFMA projection kernel -- GEMM parity tests and pipeline bit tests under ONSSEN_X3R=1, then a same-box A/B
export TMPDIR=/tmp
mkdir -p gpurun_out
ONSSEN_X3R=1 timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_pipeline.py tests/test_gpu_ragged.py -m gpu -q -x > gpurun_out/r06c5_pytest.log 2>&1; echo "pytest (X3R=1) rc $?"; tail -5 gpurun_out/r06c5_pytest.log
python tools/ab_env.py ONSSEN_X3R 0 1 -- bench.py --no-cpu-baseline --no-extra --steps 40 2>&1 | tee gpurun_out/r06c5_ab.txt
