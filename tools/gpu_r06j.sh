#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
python tools/ab_variants.py run base abl64 abl128 abl192 -- bench.py --no-cpu-baseline --no-extra --steps 40 > gpurun_out/r06j_ab.txt 2>&1; cat gpurun_out/r06j_ab.txt
