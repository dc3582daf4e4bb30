#!/bin/bash
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "loss or train" 2>&1 | tail -2
for i in 1 2; do timeout 200 python tools/train_step_bench.py --layers 3 2>&1 | tail -1 | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('L3', r.get('ms_per_step'), r.get('last_loss'))"; done
timeout 100 python tools/train_op_profile.py 2>&1 | grep -E "loss_dc|l2norm|bn_rows|dropout_kernel" | cut -c1-60,150-330 | awk '{print $1, $2, $(NF-5), $(NF-4), $NF}'
