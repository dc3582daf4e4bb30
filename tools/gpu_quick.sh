#!/bin/bash
export TMPDIR=/tmp
timeout 600 python -m pytest tests -m gpu -x -q 2>&1 | tail -2
timeout 300 python tools/separate_probe.py 2>&1 | grep -v amdgpu.ids | tail -1
