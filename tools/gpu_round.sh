#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== smoke" ; timeout 300 python -c "import __graft_entry__ as g; g.build(); g.smoke()" 2>&1 | tail -1
echo "== bench default" ; timeout 900 python bench.py 2>&1 | tail -1 | tee gpurun_out/bench.log | cut -c1-200
echo "== bench f32" ; timeout 600 python bench.py --precision f32 --no-cpu-baseline 2>&1 | tail -1 > gpurun_out/bench_f32.log
echo "== rocprof" ; cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/rocprof.log 2>&1; cd $GRAFT_REPO_ROOT
f=$(find gpurun_out/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" gpurun_out/kernel_stats.csv && cut -c1-150 "$f" | head -8
echo "== pmc" ; cd /tmp
for pass in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE"; do
  tag=$(echo $pass | cut -d' ' -f1)
  timeout 900 rocprofv3 --pmc $pass --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc_$tag -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-graph > $GRAFT_REPO_ROOT/gpurun_out/pmc_$tag.log 2>&1
done
cd $GRAFT_REPO_ROOT
python tools/pmc_summary.py gpurun_out > gpurun_out/pmc_summary.txt 2>&1; head -50 gpurun_out/pmc_summary.txt
find gpurun_out -name "*.csv" -size +3M -delete
