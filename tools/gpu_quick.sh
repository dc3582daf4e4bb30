#!/bin/bash
export TMPDIR=/tmp
for pe in 1 0 1 0; do echo "persist=$pe"; ONSSEN_X3Q_PERSIST=$pe SHAPES="[(12800,4800,1200),(12800,2580,1200),(12800,4800,129),(25600,4800,1200),(6400,4800,1200)]" timeout 100 python tools/gemm_probe.py; done
for pe in 1 0 1 0; do echo -n "bench persist=$pe: "; ONSSEN_X3Q_PERSIST=$pe timeout 100 python bench.py --no-cpu-baseline --steps 40 | python -c "import sys,json; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(r['ms_per_step'], r['roofline']['other_kernels']['ms_by_call'])"; done
timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "golden or linear or gemm or cfg" 2>&1 | tail -2
