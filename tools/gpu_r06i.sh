#!/bin/bash
# round 6, call i: whole GPU suite, round profile (kernel stats + PMC passes), default bench line
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 1800 python -m pytest tests -m gpu -q -x > gpurun_out/r06i_pytest_gpu.log 2>&1; echo "pytest rc $?"; tail -3 gpurun_out/r06i_pytest_gpu.log
bash tools/profile_round.sh r06 > gpurun_out/r06i_profile.log 2>&1; tail -30 gpurun_out/r06i_profile.log
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r06i_bench.json 2> gpurun_out/r06i_bench.err; echo "bench rc $?"; python - <<'P'
import json
r = json.loads(open('gpurun_out/r06i_bench.json').read().strip().splitlines()[-1])
print('ms_per_step', r['ms_per_step'], 'value', r['value'], 'roofline', {k: r['roofline'].get(k) for k in ('achieved', 'frac', 'us_per_time_step', 'traffic')})
c = r['cpu_baseline']; print('cpu_baseline', {k: c[k] for k in ('value', 'cores', 'cores_physical', 'threads_used', 'kmeans_n_init')}); print({k: (v['ms'], v['passes'], v['threads']) for k, v in c['whole_path'].items()}); print({k: (v['ms'], v['passes'], v['threads']) for k, v in c['network_only'].items()})
P
