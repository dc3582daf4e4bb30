#!/bin/bash
export TMPDIR=/tmp
timeout 600 python -m pytest tests -m gpu -q -x -k "cluster or ragged or separation or separate or robust" > gpurun_out/pytest_gpu_quick.log 2>&1; tail -3 gpurun_out/pytest_gpu_quick.log
timeout 300 python -c "import __graft_entry__ as g; g.build(); g.smoke()" 2>&1 | tail -1
timeout 200 python tools/cluster_probe.py 2>&1 | tail -4 | cut -c1-600
timeout 300 python bench.py --no-extra --no-cpu-baseline --steps 60 --warmup 5 > gpurun_out/bench_quick.json 2>/dev/null
python - <<'PY'
import json
r = json.loads(open("gpurun_out/bench_quick.json").read().strip().splitlines()[-1])
print("headline ms/step %.4f resident %.4f" % (r["ms_per_step"], r["resident_mask_step"]["ms_per_step"]), r["roofline"].get("dc_back_end_legs_ms"))
PY
