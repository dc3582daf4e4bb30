#!/bin/bash
# GPU box: training with layers wider than 640 -- gradient tests, then the step with the persistent forward on / off
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "training_without_the_persistent or training_gradients_match" 2>&1 | tail -6
for x in 1 0; do
  ONSSEN_XCD=$x timeout 300 python tools/train_step_bench.py --layers 2 --hidden 768 --steps 5 --warmup 2 2>/dev/null < /dev/null | tail -1 | python -c "
import json,sys; r=json.loads(sys.stdin.read()); print('train dc_l2 H=768 ONSSEN_XCD=$x ms/step %.2f loss %.2f' % (r['ms_per_step'], r['last_loss']))"
done | tee gpurun_out/wide_train.txt
