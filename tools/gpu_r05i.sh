#!/bin/bash
# round 5: the backward recurrence writes dP as the gradient GEMMs' image -- tests, A/B, kernel stats
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -q -x -k "train or grad or backward or cfg4 or loss or recipe or adam" > gpurun_out/pytest_train.log 2>&1; echo "pytest rc $?"; tail -3 gpurun_out/pytest_train.log; grep -E "^FAILED|Error" gpurun_out/pytest_train.log | head
for v in 1 0 1 0; do ONSSEN_TRAIN_DP_IMAGE=$v timeout 300 python tools/train_step_bench.py --layers 3 --steps 30 --warmup 5 2>/dev/null | tail -1 > gpurun_out/train_dpi$v.json; python -c "
import json; r=json.load(open('gpurun_out/train_dpi$v.json')); print('dp_image=$v train_step dc_l3 ms/step %.3f loss %.4f' % (r['ms_per_step'], r['last_loss']))"; done
timeout 300 python tools/train_step_bench.py --layers 2 --steps 30 --warmup 5 2>/dev/null | tail -1 > gpurun_out/train_l2.json; python -c "
import json; r=json.load(open('gpurun_out/train_l2.json')); print('train_step dc_l2 ms/step %.3f' % r['ms_per_step'])"
cd /tmp; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_train -- python $GRAFT_REPO_ROOT/tools/train_step_bench.py --layers 3 --steps 10 --warmup 3 > $GRAFT_REPO_ROOT/gpurun_out/prof_train.log 2>&1 < /dev/null; cd $GRAFT_REPO_ROOT
f=$(ls -t $(find gpurun_out/prof_train -name "*kernel_stats.csv") | head -1); if [ -n "$f" ]; then cp $f gpurun_out/train_kernel_stats.csv; head -12 $f | cut -c1-150; fi
find gpurun_out/prof_train -name "*kernel_trace.csv" -delete
