#!/bin/bash
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_robustness.py tests/test_config_surface.py -m gpu -q -x -k "train or grad or dropout or robust or loss or config or recipe or batch_norm" 2>&1 | tail -3
for i in 1 2; do timeout 200 python tools/train_step_bench.py --layers 3 2>&1 | tail -1 | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('L3', r.get('ms_per_step'), r.get('last_loss'))"; done
timeout 200 python tools/train_step_bench.py --layers 2 2>&1 | tail -1 | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('L2', r.get('ms_per_step'), r.get('last_loss'))"
