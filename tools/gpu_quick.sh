#!/bin/bash
export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "golden or xcd or cfg" 2>&1 | tail -4
python tools/ab_variants.py run fold0 fold1 -- bench.py --no-cpu-baseline --steps 40
python tools/ab_variants.py run fold0 fold1 -- bench.py --no-cpu-baseline --steps 40 --config dc_l3
