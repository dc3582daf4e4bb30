#!/bin/bash
export TMPDIR=/tmp
python tools/ab_variants.py run base perm -- bench.py --no-cpu-baseline --steps 40
timeout 300 python -m pytest tests/test_gpu_parity.py tests/test_gpu_robustness.py -m gpu -q -x -k "golden or xcd or cfg or finite or abort" 2>&1 | tail -2
