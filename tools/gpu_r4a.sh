#!/bin/bash
# round 4, first GPU pass: parity suite (incl. the ragged tests), smoke, evaluation-loop probe, bench
export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -q -x > gpurun_out/pytest_gpu.log 2>&1; tail -5 gpurun_out/pytest_gpu.log
grep -E "^\[(bf16|cfg4)" -A12 gpurun_out/pytest_gpu.log | head -60
timeout 300 python -c "import __graft_entry__ as g; g.build(); g.smoke()" 2>&1 | tail -2
timeout 300 python tools/eval_probe.py 2>&1 | grep -v Warning | tail -10 | tee gpurun_out/eval_probe.txt
timeout 500 python bench.py > gpurun_out/bench_dc_l2.json 2> gpurun_out/bench_dc_l2.err; tail -c 600 gpurun_out/bench_dc_l2.err
python - <<'PY'
import json
r = json.loads(open("gpurun_out/bench_dc_l2.json").read().strip().splitlines()[-1])
print("headline ms/step", r["ms_per_step"], "xRT", r["value"], "legs", r["roofline"].get("legs_ms"), "le", r["roofline"].get("legs_le_step"))
print("resident", r.get("resident_mask_step"))
for k, v in r.get("extra_configs", {}).items():
    print(k, {a: (round(b, 3) if isinstance(b, float) else b) for a, b in v.items() if a in ("ms_per_step", "x_real_time", "ms_per_utterance", "one_by_one_ms_per_utterance", "error", "padding_overhead")})
print("cpu", {k: v for k, v in r.get("cpu_baseline", {}).items() if k != "network_only"})
PY
