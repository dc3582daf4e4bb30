#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== boundary" ; timeout 600 python tools/boundary_probe.py 2>&1 | tail -6 | tee gpurun_out/boundary.log
