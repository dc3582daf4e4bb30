#!/bin/bash
# round 5: the bench as the driver runs it (fresh process first thing on the box), then the 2-rank self-launch
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 600 python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/bench_driver_like.json 2> gpurun_out/bench_driver_like.err; echo "bench rc $?"
ONSSEN_BENCH_ONE_DEVICE=1 timeout 600 python bench.py --gpus 2 --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/bench_selflaunch_2rank.json 2> gpurun_out/bench_selflaunch_2rank.err; echo "selflaunch rc $?"
timeout 300 python3 bench.py --gpus 1 --steps 20 --warmup 5 --preheat 0 --no-extra --no-cpu-baseline > gpurun_out/bench_nopreheat.json 2>/dev/null
python - <<'PY'
import json
for f in ("bench_driver_like", "bench_nopreheat"):
    r = json.loads(open(f"gpurun_out/{f}.json").read().strip().splitlines()[-1])
    print(f, "ms/step %.4f xRT %.0f" % (r["ms_per_step"], r["value"]), "second", (r.get("second_input_set") or {}).get("ms_per_step"), "legs_sum", r["roofline"].get("legs_sum_ms"))
r2 = json.loads(open("gpurun_out/bench_selflaunch_2rank.json").read().strip().splitlines()[-1])
print("selflaunch n_gpus", r2["n_gpus"], r2["per_rank_ms_per_step"], (r2.get("dp_training_step_dc_l3_b16") or {}).get("ms_per_step"), (r2.get("dp_training_step_dc_l3_b16") or {}).get("replicas_identical_after_dp_steps"))
PY
