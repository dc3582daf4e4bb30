#!/bin/bash
# round 5: the whole stack's training images in one launch -- tests, step time, kernel stats, wide layers
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -q -x -k "train or grad or backward or cfg4 or loss or recipe or adam or wide" > gpurun_out/pytest_train.log 2>&1; echo "pytest rc $?"; tail -3 gpurun_out/pytest_train.log; grep -E "^FAILED|Error" gpurun_out/pytest_train.log | head
for i in 1 2 3; do timeout 300 python tools/train_step_bench.py --layers 3 --steps 30 --warmup 5 2>/dev/null | tail -1 > gpurun_out/train_l3.json; python -c "
import json; r=json.load(open('gpurun_out/train_l3.json')); print('train_step dc_l3 ms/step %.3f loss %.4f' % (r['ms_per_step'], r['last_loss']))"; done
timeout 300 python tools/train_step_bench.py --layers 2 --steps 30 --warmup 5 2>/dev/null | tail -1 > gpurun_out/train_l2.json; python -c "
import json; r=json.load(open('gpurun_out/train_l2.json')); print('train_step dc_l2 ms/step %.3f' % r['ms_per_step'])"
timeout 300 python tools/train_step_bench.py --layers 2 --hidden 768 --steps 10 --warmup 3 2>/dev/null | tail -1 | python -c "
import json,sys; r=json.loads(sys.stdin.read()); print('train_step 2x768 ms/step %.3f loss %.3f' % (r['ms_per_step'], r['last_loss']))"
rm -rf gpurun_out/prof_train; cd /tmp; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_train -- python $GRAFT_REPO_ROOT/tools/train_step_bench.py --layers 3 --steps 10 --warmup 3 > $GRAFT_REPO_ROOT/gpurun_out/prof_train.log 2>&1 < /dev/null; cd $GRAFT_REPO_ROOT
f=$(ls -t $(find gpurun_out/prof_train -name "*kernel_stats.csv") | head -1); if [ -n "$f" ]; then cp $f gpurun_out/train_kernel_stats.csv; grep -E "pack|Name" $f | cut -c1-150; fi
find gpurun_out/prof_train -name "*kernel_trace.csv" -delete
