#!/bin/bash
# round 5, second GPU call: the whole -m gpu suite after the options refactor / per-stream workspaces / new loader, the loader probe,
# the bench with its trained-weights leg, the self-launching 2-rank harness test again (short DP leg).
export TMPDIR=/tmp
mkdir -p gpurun_out
rm -f gpurun_out/eval_length_errors.json
timeout 1500 python -m pytest tests -m gpu -q -s > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc $?"; tail -3 gpurun_out/pytest_gpu.log
grep -E "^FAILED|^ERROR|^1000 steps|^ragged K=" gpurun_out/pytest_gpu.log | cut -c1-300 | tail -30
timeout 300 python -c "import __graft_entry__ as g; g.build(); g.smoke()" 2>&1 | tail -2
timeout 600 python tools/loader_probe.py > gpurun_out/loader_probe.json 2> gpurun_out/loader_probe.err; echo "loader rc $?"; cat gpurun_out/loader_probe.json | cut -c1-1500; tail -3 gpurun_out/loader_probe.err
timeout 600 python bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err; echo "bench rc $?"; tail -c 400 gpurun_out/bench_default.err
ONSSEN_BENCH_ONE_DEVICE=1 timeout 600 python bench.py --gpus 2 --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/bench_selflaunch_2rank.json 2> gpurun_out/bench_selflaunch_2rank.err; echo "selflaunch rc $?"
python - <<'PY'
import json
r = json.loads(open("gpurun_out/bench_default.json").read().strip().splitlines()[-1])
print("headline ms/step %.3f xRT %.0f" % (r["ms_per_step"], r["value"]), "lloyd", r.get("lloyd_iterations", {}).get("mean"), "second", r.get("second_input_set", {}).get("ms_per_step"))
print("legs", r["roofline"].get("legs_ms"))
print("trained", json.dumps(r["extra_configs"].get("trained_weights_dc_l2_b32"))[:1200])
print("train step", r["extra_configs"]["cfg4_training_step_dc_l3_b16"].get("ms_per_step"), r["extra_configs"]["cfg4_training_step_dc_l3_b16"].get("roofline_backward_recurrence", {}).get("us_per_time_step"))
r2 = json.loads(open("gpurun_out/bench_selflaunch_2rank.json").read().strip().splitlines()[-1])
print("selflaunch n_gpus", r2["n_gpus"], r2["per_rank_ms_per_step"], json.dumps(r2.get("dp_training_step_dc_l3_b16"))[:900])
PY
