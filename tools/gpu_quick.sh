#!/bin/bash
export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "training" 2>&1 | grep -E "^E  |passed|failed|FAILED" | cut -c1-600 | head -12
timeout 200 python tools/train_step_bench.py --layers 3 --steps 10 --warmup 3 2>&1 < /dev/null | tail -1 > gpurun_out/train_x.json
timeout 20 python -c "
import json; r=json.load(open('gpurun_out/train_x.json')); print('train', r['ms_per_step'], r['value'], r['last_loss'])"
timeout 100 python bench.py --no-cpu-baseline 2>&1 | tail -1 | python -c "import json,sys; r=json.loads(sys.stdin.read()); print('dc_l2', r['ms_per_step'], r['roofline']['other_kernels']['ms_by_call'])"
