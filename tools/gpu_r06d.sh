#!/bin/bash
# round 6, call d: input projection requested early + an explicit pause before the loaders' chunk request
export TMPDIR=/tmp
mkdir -p gpurun_out
python tools/ab_variants.py run base g1 g1d4 g1d6 g1d8 g1d10 -- bench.py --no-cpu-baseline --no-extra --steps 40 > gpurun_out/r06d_ab.txt 2>&1; cat gpurun_out/r06d_ab.txt
