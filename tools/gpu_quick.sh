#!/bin/bash
export TMPDIR=/tmp
python tools/ab_variants.py run base nobr st2 st3 -- bench.py --no-cpu-baseline --steps 40
