#!/bin/bash
export TMPDIR=/tmp
python tools/ab_variants.py run base wxlds -- bench.py --no-cpu-baseline --steps 20 --config chimera_l4
for f in auto 1 auto 1; do echo -n "dc_l2 wxlds fuse=$f: "; ONSSEN_FUSE_IN0=$f ONSSEN_HIP_LIB=build_variants/libonssen_hip_wxlds.so timeout 100 python bench.py --no-cpu-baseline --steps 40 | python -c "import sys,json; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(r['ms_per_step'], r['roofline']['us_per_time_step'])"; done
for f in auto 1; do echo -n "dc_l3 wxlds fuse=$f: "; ONSSEN_FUSE_IN0=$f ONSSEN_HIP_LIB=build_variants/libonssen_hip_wxlds.so timeout 100 python bench.py --no-cpu-baseline --steps 40 --config dc_l3 | python -c "import sys,json; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(r['ms_per_step'], r['roofline']['us_per_time_step'])"; done
timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "fused or cfg3 or chimera or ragged" 2>&1 | tail -2
