#!/bin/bash
export TMPDIR=/tmp
for v in base wearly; do echo $v; ONSSEN_HIP_LIB=build_variants/libonssen_hip_$v.so timeout 100 python tools/xcd_startup_probe.py 2>&1 | tail -5; done
python tools/ab_variants.py run base wearly -- bench.py --no-cpu-baseline --steps 40
