#!/bin/bash
export TMPDIR=/tmp
for bm in 256 128 0; do echo "BM=$bm (0 = cost model)"; ONSSEN_X3Q_BM=$bm SHAPES="[(6400,1280,4800),(6400,1200,2580),(2580,1200,6400),(6400,4800,1200),(12800,4800,1200),(6400,2580,1200)]" timeout 100 python tools/gemm_probe.py; done
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "golden or linear or gemm or train or grad" 2>&1 | tail -2
for i in 1 2; do timeout 200 python tools/train_step_bench.py --layers 3 2>&1 | tail -1 | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('L3', r.get('ms_per_step'), r.get('last_loss'))"; done
ONSSEN_X3Q_BM=256 timeout 200 python tools/train_step_bench.py --layers 3 2>&1 | tail -1 | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('L3 BM=256 forced', r.get('ms_per_step'), r.get('last_loss'))"
timeout 100 python bench.py --no-cpu-baseline --steps 40 | python -c "import sys,json; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(r['ms_per_step'], r['roofline']['other_kernels']['ms_by_call'])"
timeout 100 python bench.py --no-cpu-baseline --steps 40 --config dc_l3 | python -c "import sys,json; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('dc_l3', r['ms_per_step'], r['roofline']['other_kernels']['ms_by_call'])"
