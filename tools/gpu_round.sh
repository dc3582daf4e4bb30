#!/bin/bash
# Round-end validation on the GPU box: parity suite, smoke, every bench configuration, rocprofv3 passes.
# Usage: bash tools/gpu_round.sh <tag>   (tag names the profile directory gpurun_out/prof_<tag>, e.g. r02a)
tag=${1:-r05}
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1; tail -2 gpurun_out/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.build(); g.smoke(); print('smoke ok')" 2>&1 | tail -1
run() { name=$1; shift; timeout 400 env "$@" python bench.py ${BENCH_ARGS} > gpurun_out/bench_$name.json 2> gpurun_out/bench_$name.err; python - <<PY
import json
try:
    r = json.loads(open("gpurun_out/bench_$name.json").read().strip().splitlines()[-1])
    print("$name", "ms/step %.3f" % r["ms_per_step"], "xRT %.0f" % r["value"], "fps %.0f" % r["frames_per_s"], "rec us/step %.2f" % r["roofline"]["us_per_time_step"], "frac %.3f" % r["roofline"]["frac"], "|", r["config"].get("recurrence"), "| cpu", round(r["cpu_baseline"]["value"]) if r.get("cpu_baseline") else None, r["roofline"]["other_kernels"]["ms_by_call"])
except Exception as e:
    print("$name FAILED", e, open("gpurun_out/bench_$name.err").read()[-400:])
PY
}
BENCH_ARGS="" run dc_l2_bf16x3 A=1
BENCH_ARGS="--config dc_l2 --precision f32 --no-cpu-baseline --no-extra" run dc_l2_f32 A=1
BENCH_ARGS="--config dc_l2 --no-cpu-baseline --no-extra" run dc_l2_bf16x3_steps ONSSEN_XCD=0
BENCH_ARGS="--config dc_l3 --no-cpu-baseline --no-extra" run dc_l3 A=1
BENCH_ARGS="--config chimera_l4 --no-cpu-baseline --no-extra" run chimera_l4 A=1
BENCH_ARGS="--config phase_l4 --no-cpu-baseline --no-extra" run phase_l4 A=1
BENCH_ARGS="--config dc_l2 --precision bf16 --no-cpu-baseline --no-extra" run dc_l2_bf16_optin A=1
for cfg in "1 3" "0 3" "1 2"; do
  set -- $cfg; mode=$1; layers=$2
  ONSSEN_TRAIN_HIP=$mode timeout 300 python tools/train_step_bench.py --layers $layers --steps 10 --warmup 3 2> gpurun_out/train_${mode}_l$layers.err < /dev/null | tail -1 > gpurun_out/train_hip${mode}_l$layers.json
  timeout 20 python -c "
import json; r=json.load(open('gpurun_out/train_hip${mode}_l$layers.json')); print('train_step dc_l$layers hip=$mode ms/step %.2f xRT %.0f loss %.2f' % (r['ms_per_step'], r['value'], r['last_loss']))" 2>&1 | tail -1
done
timeout 200 python tools/train_soak.py --steps 100 2>&1 < /dev/null | tail -1 | cut -c1-400 | tee gpurun_out/train_soak.txt
timeout 300 python tools/xcd_soak.py 2>&1 < /dev/null | tail -4 | cut -c1-400 | tee gpurun_out/xcd_soak.txt
cd /tmp; timeout 240 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_train -- python $GRAFT_REPO_ROOT/tools/train_step_bench.py --layers 3 --steps 10 --warmup 3 > $GRAFT_REPO_ROOT/gpurun_out/prof_train.log 2>&1 < /dev/null; cd $GRAFT_REPO_ROOT
f=$(find gpurun_out/prof_train -name "*kernel_stats.csv" | head -1); if [ -n "$f" ]; then cp $f gpurun_out/train_kernel_stats.csv; fi
find gpurun_out/prof_train -name "*kernel_trace.csv" -delete
timeout 300 python tools/eval_probe.py 2>&1 | grep -v Warning | grep "batch" | tee gpurun_out/eval_probe.txt
HEAVY=1 timeout 200 python tools/cotenant_probe.py > gpurun_out/cotenant_probe.txt 2>/dev/null; tail -12 gpurun_out/cotenant_probe.txt
timeout 100 python tools/cluster_probe.py 2>/dev/null | head -3 | cut -c1-200 | tee gpurun_out/cluster_probe.txt
timeout 200 python tools/wide_layer_probe.py 2>/dev/null | tail -1 | tee gpurun_out/wide_probe.json
rm -rf gpurun_out/prof_wide; cd /tmp; WIDE_ONLY=768 timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_wide -- python $GRAFT_REPO_ROOT/tools/wide_layer_probe.py > $GRAFT_REPO_ROOT/gpurun_out/prof_wide.log 2>&1 < /dev/null; cd $GRAFT_REPO_ROOT
f=$(find gpurun_out/prof_wide -name "*kernel_stats.csv" | head -1); if [ -n "$f" ]; then head -6 $f | cut -c1-160 | tee gpurun_out/wide_kernel_stats.txt; fi
find gpurun_out/prof_wide -name "*kernel_trace.csv" -delete
bash tools/profile_round.sh $tag > gpurun_out/profile_round.log 2>&1
head -9 gpurun_out/prof_$tag/kernel_stats.csv | cut -c1-170
