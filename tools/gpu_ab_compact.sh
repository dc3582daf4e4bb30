#!/bin/bash
export TMPDIR=/tmp
for rep in 1; do
for c in 1 0; do
  ONSSEN_DC_COMPACT=$c timeout 300 python bench.py --no-extra --no-cpu-baseline --steps 60 --warmup 5 > gpurun_out/ab_$c.json 2>/dev/null
  python - <<PY
import json
r = json.loads(open("gpurun_out/ab_$c.json").read().strip().splitlines()[-1])
print("compact=$c headline ms/step %.4f resident %.4f" % (r["ms_per_step"], r["resident_mask_step"]["ms_per_step"]), r["roofline"].get("dc_back_end_legs_ms"), r["roofline"]["legs_ms"]["threshold_2means_masks"])
PY
done
done
cd /tmp; timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_ab -- python $GRAFT_REPO_ROOT/bench.py --no-extra --no-cpu-baseline --no-graph --steps 10 --warmup 2 > /dev/null 2>&1
cd $GRAFT_REPO_ROOT; f=$(find gpurun_out/prof_ab -name "*kernel_stats.csv" | head -1); head -30 $f | cut -c1-150; find gpurun_out/prof_ab -name "*kernel_trace.csv" -delete
