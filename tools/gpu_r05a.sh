#!/bin/bash
# round 5, first GPU call: full -m gpu suite (new: evaluation-length parity, trained-embedding probe), the bench with its batch sweep,
# the self-launching 2-rank harness test on one device, the long trained probe, the ug = 24 A/B of the recurrence.
export TMPDIR=/tmp
mkdir -p gpurun_out
rm -f gpurun_out/eval_length_errors.json
timeout 1200 python -m pytest tests -m gpu -x -q -s > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc $?"; tail -3 gpurun_out/pytest_gpu.log
grep -E "^dc L=|^ragged K=|^800 steps|agree" gpurun_out/pytest_gpu.log | cut -c1-260 | tail -40
timeout 600 python bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err; echo "bench rc $?"; tail -c 600 gpurun_out/bench_default.err
ONSSEN_BENCH_ONE_DEVICE=1 timeout 600 python bench.py --gpus 2 --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/bench_selflaunch_2rank.json 2> gpurun_out/bench_selflaunch_2rank.err; echo "selflaunch rc $?"; tail -c 1500 gpurun_out/bench_selflaunch_2rank.err; tail -c 2500 gpurun_out/bench_selflaunch_2rank.json
timeout 600 python tools/trained_probe.py --steps 2000 --checkpoints 0,500,1000 > gpurun_out/trained_probe.txt 2> gpurun_out/trained_probe.err; echo "probe rc $?"; grep -E "^after|^trained" gpurun_out/trained_probe.txt | cut -c1-400; tail -5 gpurun_out/trained_probe.err
for b in 32 64; do
  for ug in 0 24; do
    ONSSEN_XCD_UG=$ug ONSSEN_FUSE_IN0=0 timeout 300 python bench.py --batch $b --no-extra --no-cpu-baseline > gpurun_out/bench_ug${ug}_b$b.json 2> gpurun_out/bench_ug${ug}_b$b.err
    python - <<PY
import json
try:
    r = json.loads(open("gpurun_out/bench_ug${ug}_b$b.json").read().strip().splitlines()[-1])
    print("ug=$ug B=$b: ms/step %.3f  rec us/step %.3f frac %.3f unit_group %s" % (r["ms_per_step"], r["roofline"]["us_per_time_step"], r["roofline"]["frac"], r["roofline"]["unit_group"]))
except Exception as e:
    print("ug=$ug B=$b FAILED", e, open("gpurun_out/bench_ug${ug}_b$b.err").read()[-600:])
PY
  done
done
python - <<'PY'
import json
r = json.loads(open("gpurun_out/bench_default.json").read().strip().splitlines()[-1])
print("headline ms/step %.3f xRT %.0f" % (r["ms_per_step"], r["value"]), "lloyd", r.get("lloyd_iterations", {}).get("mean"), "second", r.get("second_input_set", {}).get("ms_per_step"))
print("legs", r["roofline"].get("legs_ms"))
for k, v in r.get("extra_configs", {}).items():
    if k == "batch_sweep":
        for cfg, s in v.items():
            if isinstance(s, dict):
                print(cfg, "knee", s.get("knee_chunks"))
                for row in s["rows"]:
                    print("   ", {kk: (round(vv, 3) if isinstance(vv, float) else vv) for kk, vv in row.items()})
    else:
        print(k, v.get("ms_per_step"), v.get("x_real_time"), v.get("error"))
PY
