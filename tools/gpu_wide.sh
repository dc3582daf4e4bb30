#!/bin/bash
# GPU box: the wide-layer persistent form -- tests, then the probe
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -s -k "wide_layers or without_the_persistent or xcd_local_persistent" 2>&1 | tail -25 > gpurun_out/wide_tests.log
cat gpurun_out/wide_tests.log
timeout 300 python tools/wide_layer_probe.py 2>&1 | tail -3 | tee gpurun_out/wide_probe.json
