#!/bin/bash
# Scratch wrapper for one gpurun call while iterating (edit freely).
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "golden or cfg3 or chimera or xcd or stft" 2>&1 | tail -3
for r in 1 2; do for f in auto 1; do echo "dc_l2 FUSE_IN0=$f"; ONSSEN_FUSE_IN0=$f timeout 200 python bench.py --no-cpu-baseline --no-extra --steps 40 2>/dev/null | python -c "import json,sys; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(r['ms_per_step'])"; done; done
for f in auto 0; do echo "chimera_l4 FUSE_IN0=$f"; ONSSEN_FUSE_IN0=$f timeout 200 python bench.py --config chimera_l4 --no-cpu-baseline --no-extra --steps 20 2>/dev/null | python -c "import json,sys; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(r['ms_per_step'])"; done
for f in auto 1; do echo "dc_l3 b16 FUSE_IN0=$f"; ONSSEN_FUSE_IN0=$f timeout 200 python bench.py --config dc_l3 --no-cpu-baseline --no-extra --steps 20 2>/dev/null | python -c "import json,sys; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(r['ms_per_step'])"; done
