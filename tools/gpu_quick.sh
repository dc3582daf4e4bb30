#!/bin/bash
export TMPDIR=/tmp
timeout 400 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -s -k "optin_bf16 or plain_bf16" 2>&1 | grep -v "^$" | tail -12
for c in dc_l2 chimera_l4; do
  timeout 300 python bench.py --config $c --precision bf16 --no-cpu-baseline 2>&1 | tail -1 > gpurun_out/bench_bf16_$c.json
  python - <<PY
import json
r=json.load(open("gpurun_out/bench_bf16_$c.json"))
print("$c", r["ms_per_step"], r["value"], r["roofline"]["us_per_time_step"], r["roofline"]["frac"], r["roofline"]["other_kernels"]["ms_by_call"])
PY
done
timeout 200 python bench.py --config dc_l2 --no-cpu-baseline 2>&1 | tail -1 | python -c "import json,sys; r=json.loads(sys.stdin.read()); print('x3', r['ms_per_step'], r['roofline']['us_per_time_step'], r['roofline']['other_kernels']['ms_by_call'])"
