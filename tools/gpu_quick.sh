#!/bin/bash
export TMPDIR=/tmp
ONSSEN_BENCH_ONE_DEVICE=1 ONSSEN_XCD=0 timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 5 --warmup 2 2>&1 | grep -v amdgpu.ids | tail -3 | cut -c1-600
