#!/bin/bash
export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "stft or istft or round or front or back or separate" 2>&1 | tail -2
timeout 120 python tools/fft_probe.py 2>&1 | grep -v amdgpu.ids | tail -3
