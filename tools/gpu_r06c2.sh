#!/bin/bash
# round 6c, call 2: the whole GPU suite on the ABI-14 tree, then the default bench line
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 2400 python -m pytest tests -m gpu -q -x > gpurun_out/r06c2_pytest_gpu.log 2>&1; echo "pytest rc $?"; tail -4 gpurun_out/r06c2_pytest_gpu.log
timeout 1200 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r06c2_bench.json 2> gpurun_out/r06c2_bench.err; echo "bench rc $?"; python - <<'P'
import json
r = json.loads(open('gpurun_out/r06c2_bench.json').read().strip().splitlines()[-1])
print('ms_per_step', r['ms_per_step'], 'value', r['value'], 'roofline', {k: r['roofline'].get(k) for k in ('achieved', 'frac', 'us_per_time_step', 'traffic', 'legs_sum_ms', 'legs_le_step')})
print('one batch', r['one_batch_at_a_time_step']['ms_per_step'], 'second', r['second_input_set']['ms_per_step'])
ex = r['extra_configs']; print('ragged', json.dumps(ex['b16_ragged_utterances'])[:1500])
print('train', ex['cfg4_training_step_dc_l3_b16'].get('ms_per_step'))
P
