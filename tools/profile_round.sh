#!/bin/bash
# GPU box: rocprofv3 kernel stats + PMC passes of the default bench workload (dc_l2), one counter group per pass.
# Usage: bash tools/profile_round.sh <tag>     -> gpurun_out/prof_<tag>/
tag=${1:-r01}
root=$PWD/gpurun_out/prof_$tag
rm -rf $root      # (gpurun merges into gpurun_out/: a stale pass of an earlier call would be averaged in)
mkdir -p $root
export TMPDIR=/tmp
cd /tmp
B="python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-graph --no-cpu-baseline --no-extra"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $root/stats -- $B > $root/stats.log 2>&1
P="python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --no-graph --no-cpu-baseline --no-extra"
i=0
for c in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_ANY" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $root/pmc_$i -- $P > $root/pmc_$i.log 2>&1
done
cd $GRAFT_REPO_ROOT
find $root/stats -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $root/kernel_stats.csv
python tools/pmc_summary.py $root > $root/pmc_summary.txt 2>&1
python tools/traffic_from_pmc.py $root $tag > $root/traffic_latest.json 2>&1      # -> $root/traffic.json (copy to profiles/traffic.json)
# drop the bulky raw traces, keep the summaries
find $root -name "*kernel_trace.csv" -delete; find $root -name "*counter_collection.csv" -size +8M -delete
head -12 $root/kernel_stats.csv; head -60 $root/pmc_summary.txt
