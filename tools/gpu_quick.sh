#!/bin/bash
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_config_surface.py -m gpu -q -x -k "loss_dc or training_step or config or recipe" 2>&1 | tail -5
for v in 1 0 1 0; do ONSSEN_LOSS_HIP=$v timeout 200 python tools/train_step_bench.py --layers 3 2>&1 | tail -1 | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('L3 loss_hip=$v', r.get('ms_per_step'), r.get('last_loss'))"; done
