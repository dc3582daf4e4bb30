#!/bin/bash
# GPU box: the training path -- gradient / loss tests, then the cfg4 training step (timed, and under rocprofv3)
mkdir -p gpurun_out; rm -rf gpurun_out/prof_train
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "training or loss or gradient or train_mode or hip_vs_aten" 2>&1 | tail -12 > gpurun_out/train_tests.log
cat gpurun_out/train_tests.log
ONSSEN_TRAIN_HIP=1 timeout 300 python tools/train_step_bench.py --layers 3 --steps 20 --warmup 5 2> gpurun_out/train_1_l3.err < /dev/null | tail -1 > gpurun_out/train_hip1_l3.json
python -c "
import json; r=json.load(open('gpurun_out/train_hip1_l3.json')); print('train_step dc_l3 ms/step %.3f xRT %.0f loss %.2f' % (r['ms_per_step'], r['value'], r['last_loss']))" 2>&1 | tail -1
cd /tmp; export TMPDIR=/tmp; timeout 240 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_train -- python $GRAFT_REPO_ROOT/tools/train_step_bench.py --layers 3 --steps 10 --warmup 3 > $GRAFT_REPO_ROOT/gpurun_out/prof_train.log 2>&1 < /dev/null; cd $GRAFT_REPO_ROOT
f=$(find gpurun_out/prof_train -name "*kernel_stats.csv" | head -1); if [ -n "$f" ]; then cp $f gpurun_out/train_kernel_stats.csv; fi
find gpurun_out/prof_train -name "*kernel_trace.csv" -delete
head -32 gpurun_out/train_kernel_stats.csv | cut -c1-150
