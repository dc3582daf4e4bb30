#!/bin/bash
# round 5, fourth GPU call: after the stale-image fix (fused Adam does not bump parameter versions): diagnostic, training tests, probe, step time
export TMPDIR=/tmp
mkdir -p gpurun_out
STEPS=300 timeout 300 python tools/micro/train_then_eval_diag.py 2>&1 | grep -v Warn | tail -6
timeout 900 python -m pytest tests -m gpu -q -x -s -k "train or grad or backward or cfg4 or loss or recipe" > gpurun_out/pytest_train.log 2>&1; echo "pytest rc $?"; tail -3 gpurun_out/pytest_train.log; grep -E "^1000 steps|^FAILED|Error" gpurun_out/pytest_train.log | head
for l in 3 2; do timeout 300 python tools/train_step_bench.py --layers $l --steps 20 --warmup 5 2>/dev/null | tail -1 > gpurun_out/train_l$l.json; python -c "
import json; r=json.load(open('gpurun_out/train_l$l.json')); print('train_step dc_l$l ms/step %.3f loss %.2f' % (r['ms_per_step'], r['last_loss']))"; done
timeout 600 python tools/trained_probe.py --steps 2000 --checkpoints 0,250,500,1000 --routes > gpurun_out/trained_probe.txt 2> gpurun_out/trained_probe.err; echo "probe rc $?"; grep -E "route|^after|^trained" gpurun_out/trained_probe.txt | cut -c1-330; tail -3 gpurun_out/trained_probe.err
cd /tmp; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_train -- python $GRAFT_REPO_ROOT/tools/train_step_bench.py --layers 3 --steps 10 --warmup 3 > $GRAFT_REPO_ROOT/gpurun_out/prof_train.log 2>&1 < /dev/null; cd $GRAFT_REPO_ROOT
f=$(find gpurun_out/prof_train -name "*kernel_stats.csv" | head -1); if [ -n "$f" ]; then cp $f gpurun_out/train_kernel_stats.csv; head -30 $f | cut -c1-150; fi
find gpurun_out/prof_train -name "*kernel_trace.csv" -delete
