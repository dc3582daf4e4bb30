#!/bin/bash
# round 5, third GPU call: the stacked backward recurrence (gradient tests, training step), the inference-route diagnostic on a trained network
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -q -x -k "train or grad or backward or cfg4 or loss" > gpurun_out/pytest_train.log 2>&1; echo "pytest rc $?"; tail -3 gpurun_out/pytest_train.log
for l in 3 2; do timeout 300 python tools/train_step_bench.py --layers $l --steps 20 --warmup 5 2>/dev/null | tail -1 > gpurun_out/train_l$l.json; python -c "
import json; r=json.load(open('gpurun_out/train_l$l.json')); print('train_step dc_l$l ms/step %.3f loss %.2f' % (r['ms_per_step'], r['last_loss']))"; done
timeout 600 python tools/trained_probe.py --steps 1000 --checkpoints "" --eval 4 --routes > gpurun_out/trained_routes.txt 2> gpurun_out/trained_routes.err; echo "routes rc $?"; grep -E "route|^after" gpurun_out/trained_routes.txt | cut -c1-300; tail -3 gpurun_out/trained_routes.err
timeout 600 python bench.py --no-cpu-baseline > gpurun_out/bench_c.json 2> gpurun_out/bench_c.err; echo "bench rc $?"
python - <<'PY'
import json
r = json.loads(open("gpurun_out/bench_c.json").read().strip().splitlines()[-1])
t = r["extra_configs"]["cfg4_training_step_dc_l3_b16"]
print("headline", r["ms_per_step"], "train step", t.get("ms_per_step"), "bwd us/step", t.get("roofline_backward_recurrence", {}).get("us_per_time_step"), "frac", t.get("roofline_backward_recurrence", {}).get("frac"), "fwd layer ms", t.get("forward_layer_with_saved_state_ms"))
print("trained", json.dumps(r["extra_configs"].get("trained_weights_dc_l2_b32"))[:600])
PY
