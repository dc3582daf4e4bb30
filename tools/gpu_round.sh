#!/bin/bash
# One gpurun round trip: smoke, GPU parity tests, bench, ablations, PMC counters.
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== smoke" ; timeout 300 python -c "import __graft_entry__ as g; g.build(); g.smoke()" 2>&1 | tail -3
echo "== pytest gpu" ; timeout 1200 python -m pytest tests -m gpu -q --no-header -rf > gpurun_out/pytest_gpu.log 2>&1 ; tail -6 gpurun_out/pytest_gpu.log
echo "== ablate" ; timeout 600 python tools/ablate_step.py 2>&1 | tail -8 | tee gpurun_out/ablate.log
B=16 timeout 600 python tools/ablate_step.py 2>&1 | tail -8 | tee -a gpurun_out/ablate.log
echo "== pmc" ; cd /tmp
for pass in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum" "SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE"; do
  tag=$(echo $pass | cut -d' ' -f1)
  timeout 900 rocprofv3 --pmc $pass --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc_$tag -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-graph > $GRAFT_REPO_ROOT/gpurun_out/pmc_$tag.log 2>&1
done
cd $GRAFT_REPO_ROOT
python tools/pmc_summary.py gpurun_out 2>&1 | tail -40 | tee gpurun_out/pmc_summary.txt
find gpurun_out -name "*.csv" -size +3M -delete
