#!/bin/bash
# round 5, sixth GPU call: the target map on a side stream (A/B), separation tests, the round's rocprofv3 kernel stats + PMC passes
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -q -x -k "dc or separat or ragged or cluster or robust or abort or stream" > gpurun_out/pytest_sep.log 2>&1; echo "pytest rc $?"; tail -2 gpurun_out/pytest_sep.log
for ov in 0 1 0 1; do
  ONSSEN_DC_INDEX_OVERLAP=$ov timeout 300 python bench.py --no-extra --no-cpu-baseline --steps 50 --warmup 5 > gpurun_out/bench_ov$ov.json 2> gpurun_out/bench_ov$ov.err
  python - <<PY
import json
try:
    r = json.loads(open("gpurun_out/bench_ov$ov.json").read().strip().splitlines()[-1])
    print("dc_index_overlap=$ov: ms/step %.4f xRT %.0f placement-independent protocol used: %s" % (r["ms_per_step"], r["value"], r["config"]["xcd_placement_independent_protocol_used"]))
except Exception as e:
    print("overlap=$ov FAILED", e, open("gpurun_out/bench_ov$ov.err").read()[-500:])
PY
done
bash tools/profile_round.sh r05 > gpurun_out/profile_round.log 2>&1; tail -75 gpurun_out/profile_round.log | cut -c1-200
