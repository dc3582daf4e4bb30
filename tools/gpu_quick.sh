#!/bin/bash
export TMPDIR=/tmp
ONSSEN_HIP_LIB=$GRAFT_REPO_ROOT/build_variants/libonssen_hip_paired.so timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "golden or xcd_local or placement or training_gradients or graph" 2>&1 | tail -3
timeout 600 python tools/ab_variants.py run base paired -- bench.py --no-cpu-baseline --no-extra --steps 40 2>&1 | tail -5
timeout 600 python tools/ab_variants.py run base paired -- bench.py --config dc_l3 --no-cpu-baseline --no-extra --steps 40 2>&1 | tail -5
