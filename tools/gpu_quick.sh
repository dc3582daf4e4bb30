#!/bin/bash
export TMPDIR=/tmp
python tools/ab_variants.py run base pipe -- bench.py --no-cpu-baseline --steps 40
python tools/ab_variants.py run base pipe -- bench.py --no-cpu-baseline --steps 20 --config chimera_l4
python tools/ab_variants.py run base pipe -- bench.py --no-cpu-baseline --steps 40 --config dc_l3
timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "golden or xcd or cfg" 2>&1 | tail -2
