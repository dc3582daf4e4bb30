#!/bin/bash
# round 6, call n: the two-batch pipeline (DCPipeline on onssen_blstm_pipe2_forward_f32) -- parity with the 64-row route, first timing
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_pipeline.py -m gpu -x -q -s 2>&1 | grep -a -v "^  File\|^Extension" > gpurun_out/r06n_pipeline_tests.txt
head -50 gpurun_out/r06n_pipeline_tests.txt
