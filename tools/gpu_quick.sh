#!/bin/bash
export TMPDIR=/tmp
for gn in 4 5 8 15; do echo "GN=$gn"; ONSSEN_X3_GN=$gn SHAPES="[(12800,4800,1200),(12800,2580,1200),(12800,4800,129),(25600,4800,1200)]" timeout 100 python tools/gemm_probe.py; done
cd /tmp
for gn in 4 8; do
ONSSEN_X3_GN=$gn SHAPES="[(12800,4800,1200)]" timeout 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc_gn$gn -- python $GRAFT_REPO_ROOT/tools/gemm_probe.py > /dev/null 2>&1
python - <<PY
import csv,glob
f=glob.glob("$GRAFT_REPO_ROOT/gpurun_out/pmc_gn$gn/*/*counter_collection.csv")[0]
v=[float(r['Counter_Value']) for r in csv.DictReader(open(f)) if 'linear_x3q' in r['Kernel_Name']]
print("GN=$gn FETCH_SIZE KB per call", sorted(set(round(x) for x in v))[:5], len(v))
PY
done
rm -rf $GRAFT_REPO_ROOT/gpurun_out/pmc_gn*
