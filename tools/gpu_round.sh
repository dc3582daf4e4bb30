#!/bin/bash
export TMPDIR=/tmp
timeout 600 python tools/xcd_soak.py 2>&1 | grep -v amdgpu.ids | tail -4
