#!/bin/bash
export TMPDIR=/tmp
timeout 300 python tools/cumask_xcd_probe.py 2>&1 | grep -v amdgpu.ids | tail -8
