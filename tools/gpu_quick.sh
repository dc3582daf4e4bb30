#!/bin/bash
export TMPDIR=/tmp
python tools/ab_variants.py run base nobr2 -- bench.py --no-cpu-baseline --steps 40
python tools/ab_variants.py run base nobr2 -- bench.py --no-cpu-baseline --steps 20 --config chimera_l4
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_robustness.py -m gpu -q -x 2>&1 | tail -2
