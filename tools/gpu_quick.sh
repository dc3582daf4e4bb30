#!/bin/bash
# Scratch wrapper for one gpurun call while iterating (edit freely).
export TMPDIR=/tmp
timeout 300 python tools/cluster_probe.py 2>&1 | grep -v amdgpu.ids
timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "dc_cluster or e2e or separ" 2>&1 | tail -3
cd /tmp; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_cluster -- python $GRAFT_REPO_ROOT/tools/cluster_probe.py > $GRAFT_REPO_ROOT/gpurun_out/prof_cluster.log 2>&1; cd $GRAFT_REPO_ROOT
f=$(find gpurun_out/prof_cluster -name "*kernel_stats.csv" | head -1); grep -i "kmeans\|Name" $f | cut -c1-200
find gpurun_out/prof_cluster -name "*kernel_trace.csv" -delete
