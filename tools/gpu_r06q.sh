#!/bin/bash
# round 6, call q: bench with the pipelined headline + the whole GPU suite on the kernel with the pair fields
export TMPDIR=/tmp
mkdir -p gpurun_out
python bench.py --no-extra --no-cpu-baseline > gpurun_out/r06q_bench.json 2> gpurun_out/r06q_bench.err
tail -c 600 gpurun_out/r06q_bench.err
python - <<'P'
import json
r = json.loads(open("gpurun_out/r06q_bench.json").read().strip().splitlines()[-1])
print("ms_per_step", r["ms_per_step"], "value", r["value"])
print("one batch at a time", r.get("one_batch_at_a_time_step", {}).get("ms_per_step"))
roof = r["roofline"]
print({k: roof[k] for k in ("kernel", "achieved", "frac", "us_per_time_step", "us_per_launch")})
print("legs", roof["legs_ms"], roof.get("legs_sum_ms"), roof.get("legs_le_step"))
print("seq", {k: roof["one_batch_at_a_time_form"][k] for k in ("frac", "us_per_time_step")})
print("second", r.get("second_input_set"))
print("lloyd", r.get("lloyd_iterations", {}).get("mean"))
P
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -5 > gpurun_out/r06q_pytest_gpu.txt
cat gpurun_out/r06q_pytest_gpu.txt
