#!/bin/bash
# round 4: two-process harness self-test on one GPU (launch-per-step forms: two persistent launches on one device would starve each other)
export TMPDIR=/tmp
ONSSEN_BENCH_ONE_DEVICE=1 ONSSEN_XCD=0 ONSSEN_DC_PERSISTENT=0 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 \
  --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 5 --warmup 2 --no-extra --no-cpu-baseline \
  > gpurun_out/two_process_selftest.json 2> gpurun_out/two_process_selftest.err
echo rc $?; tail -c 1500 gpurun_out/two_process_selftest.json; tail -5 gpurun_out/two_process_selftest.err
