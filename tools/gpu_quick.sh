#!/bin/bash
export TMPDIR=/tmp
ONSSEN_FUSE_IN0=0 timeout 900 python tools/ab_variants.py run base batch1 -- bench.py --no-cpu-baseline --no-extra --steps 40 2>&1 | tail -4
ONSSEN_HIP_LIB=$GRAFT_REPO_ROOT/build_variants/libonssen_hip_batch1.so timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "xcd_local or determin or graph" 2>&1 | tail -2
