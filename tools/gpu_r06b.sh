#!/bin/bash
# round 6, call b: tile-group MFMA order (early partial-sum writes) A/B + cell-update ablations, same box
export TMPDIR=/tmp
mkdir -p gpurun_out
python tools/ab_bits.py base tg221 tg32 > gpurun_out/r06b_bits.txt 2>&1; tail -4 gpurun_out/r06b_bits.txt
python tools/ab_variants.py run base tg221 tg32 abl64 abl128 abl192 -- bench.py --no-cpu-baseline --no-extra --steps 40 > gpurun_out/r06b_ab.txt 2>&1; cat gpurun_out/r06b_ab.txt
