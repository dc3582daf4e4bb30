#!/bin/bash
# GPU box: PMC passes of the training step (tools/train_step_bench.py, dc_l3, 16 x 400 frames), one counter group per pass.
# Usage: bash tools/profile_train.sh <tag>   -> gpurun_out/prof_train_<tag>/
tag=${1:-r05}
root=$PWD/gpurun_out/prof_train_$tag
rm -rf $root; mkdir -p $root
export TMPDIR=/tmp
cd /tmp
P="python $GRAFT_REPO_ROOT/tools/train_step_bench.py --layers 3 --steps 2 --warmup 1"
i=0
for c in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_ANY"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $root/pmc_$i -- $P > $root/pmc_$i.log 2>&1 < /dev/null
done
cd $GRAFT_REPO_ROOT
python tools/pmc_summary.py $root > $root/pmc_summary.txt 2>&1
find $root -name "*kernel_trace.csv" -delete; find $root -name "*counter_collection.csv" -size +8M -delete
grep -A9 "lstm_xcd_bwd_kernel\|linear_x3t_kernel\|lstm_xcd_kernel" $root/pmc_summary.txt | head -60
