#!/bin/bash
export TMPDIR=/tmp
q() { python -c "import json,sys; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', round(r['ms_per_step'],4), round(r['roofline']['us_per_time_step'],3), r['roofline']['other_kernels']['ms_by_call'])"; }
for v in lds2 tmaj lds2 tmaj; do ONSSEN_HIP_LIB=$PWD/build_variants/libonssen_hip_$v.so timeout 100 python bench.py --no-cpu-baseline 2>&1 | q $v; done
timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "golden or linear or gemm" 2>&1 | tail -2
