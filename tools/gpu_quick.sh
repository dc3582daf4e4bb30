#!/bin/bash
export TMPDIR=/tmp
python tools/ab_variants.py run cur sleep4 sleep16 prio3 prio3s4 -- bench.py --no-cpu-baseline --steps 20 2>&1 | tail -10
timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "xcd or golden" 2>&1 | tail -2
