#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
python tools/ab_bits.py base skip g1skip > gpurun_out/r06f_bits.txt 2>&1; tail -3 gpurun_out/r06f_bits.txt
python tools/ab_variants.py run base skip g1 g1c1 g1c2 g1skip -- bench.py --no-cpu-baseline --no-extra --steps 40 > gpurun_out/r06f_ab.txt 2>&1; cat gpurun_out/r06f_ab.txt
