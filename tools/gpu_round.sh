#!/bin/bash
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1; tail -3 gpurun_out/pytest_gpu.log
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_fft -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-graph --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/prof_fft.log 2>&1
cd $GRAFT_REPO_ROOT
find gpurun_out/prof_fft -name "*kernel_stats.csv" | head -1 | xargs -I{} head -8 {} | cut -c1-160
find gpurun_out/prof_fft -name "*kernel_trace.csv" -delete
for c in dc_l2 phase_l4; do
timeout 300 python bench.py --config $c --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/bench_$c.json 2> gpurun_out/bench_$c.err
python - <<PY
import json
r = json.loads(open("gpurun_out/bench_$c.json").read().strip().splitlines()[-1])
print("$c", "ms/step", r["ms_per_step"], "xRT", r["value"], "rec us/step", r["roofline"].get("us_per_time_step"))
PY
done
