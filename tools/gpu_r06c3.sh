#!/bin/bash
# round 6c, call 3: final state -- smoke, whole GPU suite, round profile (kernel stats + PMC passes), default bench line
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r06c3_smoke.log 2>&1; echo "smoke rc $?"; tail -3 gpurun_out/r06c3_smoke.log
timeout 2400 python -m pytest tests -m gpu -q -x > gpurun_out/r06c3_pytest_gpu.log 2>&1; echo "pytest rc $?"; tail -3 gpurun_out/r06c3_pytest_gpu.log
bash tools/profile_round.sh r06c > gpurun_out/r06c3_profile.log 2>&1; tail -30 gpurun_out/r06c3_profile.log
timeout 1200 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r06c3_bench.json 2> gpurun_out/r06c3_bench.err; echo "bench rc $?"; python - <<'P'
import json
r = json.loads(open('gpurun_out/r06c3_bench.json').read().strip().splitlines()[-1])
print('ms_per_step', r['ms_per_step'], 'value', r['value'], 'roofline', {k: r['roofline'].get(k) for k in ('achieved', 'frac', 'us_per_time_step', 'traffic', 'legs_sum_ms', 'legs_le_step')})
print('one batch', r['one_batch_at_a_time_step']['ms_per_step'], 'second', r['second_input_set']['ms_per_step'])
c = r['cpu_baseline']; print('cpu_baseline', {k: c[k] for k in ('value', 'cores', 'cores_physical', 'threads_used', 'kmeans_n_init')})
ex = r['extra_configs']; print('trained', ex['trained_weights_dc_l2_b32'].get('ms_per_step'), ex['trained_weights_dc_l2_b32'].get('pipelined'))
print('sweep', [(row['chunks'], round(row['x_real_time'])) for row in ex['batch_sweep']['dc_l2']['rows']])
print('train', ex['cfg4_training_step_dc_l3_b16'].get('ms_per_step'))
rg = ex['b16_ragged_utterances']; print('ragged', rg.get('x_real_time'), rg.get('bucketed_by_length', {}).get('x_real_time'), rg.get('pipelined'), rg.get('plain_call_on_2K_rows'))
P
