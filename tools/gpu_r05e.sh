#!/bin/bash
# round 5, fifth GPU call: everything after the stale-image fix + lean per-step pack: full suite, training step, bench, probe (recorded), self-launch
export TMPDIR=/tmp
mkdir -p gpurun_out
rm -f gpurun_out/eval_length_errors.json
timeout 1500 python -m pytest tests -m gpu -q -s > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc $?"; tail -3 gpurun_out/pytest_gpu.log; grep -E "^FAILED|^ERROR|^1000 steps" gpurun_out/pytest_gpu.log | cut -c1-300 | head
for l in 3 2; do timeout 300 python tools/train_step_bench.py --layers $l --steps 20 --warmup 5 2>/dev/null | tail -1 > gpurun_out/train_l$l.json; python -c "
import json; r=json.load(open('gpurun_out/train_l$l.json')); print('train_step dc_l$l ms/step %.3f loss %.2f' % (r['ms_per_step'], r['last_loss']))"; done
timeout 600 python bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err; echo "bench rc $?"; tail -c 300 gpurun_out/bench_default.err
timeout 600 python tools/trained_probe.py --steps 2000 --checkpoints 0,250,500,1000 --routes > gpurun_out/trained_probe.txt 2> gpurun_out/trained_probe.err; echo "probe rc $?"; grep -E "^after" gpurun_out/trained_probe.txt | cut -c1-330
ONSSEN_BENCH_ONE_DEVICE=1 timeout 600 python bench.py --gpus 2 --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/bench_selflaunch_2rank.json 2> gpurun_out/bench_selflaunch_2rank.err; echo "selflaunch rc $?"
timeout 300 python tools/loader_probe.py > gpurun_out/loader_probe.json 2> gpurun_out/loader_probe.err; echo "loader rc $?"; cut -c1-900 gpurun_out/loader_probe.json
python - <<'PY'
import json
r = json.loads(open("gpurun_out/bench_default.json").read().strip().splitlines()[-1])
t = r["extra_configs"]["cfg4_training_step_dc_l3_b16"]
print("headline ms/step %.3f xRT %.0f" % (r["ms_per_step"], r["value"]), "lloyd", r.get("lloyd_iterations", {}).get("mean"))
print("legs", r["roofline"].get("legs_ms"))
print("train step", t.get("ms_per_step"), "bwd us/step", t.get("roofline_backward_recurrence", {}).get("us_per_time_step"), "frac", t.get("roofline_backward_recurrence", {}).get("frac"))
print("trained", json.dumps(r["extra_configs"].get("trained_weights_dc_l2_b32"))[:1100])
r2 = json.loads(open("gpurun_out/bench_selflaunch_2rank.json").read().strip().splitlines()[-1])
print("selflaunch n_gpus", r2["n_gpus"], r2["per_rank_ms_per_step"], json.dumps(r2.get("dp_training_step_dc_l3_b16"))[:600])
PY
