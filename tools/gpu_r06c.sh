#!/bin/bash
# round 6, call c: where the loader waves request the next step's input projection / h chunks
export TMPDIR=/tmp
mkdir -p gpurun_out
python tools/ab_bits.py base g1 f1 g1f1 g1f1r1 > gpurun_out/r06c_bits.txt 2>&1; tail -5 gpurun_out/r06c_bits.txt
python tools/ab_variants.py run base g1 f1 g1f1 g1f1r1 abl192 -- bench.py --no-cpu-baseline --no-extra --steps 40 > gpurun_out/r06c_ab.txt 2>&1; cat gpurun_out/r06c_ab.txt
