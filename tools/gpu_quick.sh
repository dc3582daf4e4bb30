#!/bin/bash
export TMPDIR=/tmp
python tools/ab_variants.py run base stft2 -- bench.py --no-cpu-baseline --steps 40
python tools/ab_variants.py run base stft2 -- bench.py --no-cpu-baseline --steps 20 --config phase_l4
timeout 400 python -m pytest tests/test_gpu_parity.py tests/test_wav_loader.py -m gpu -q -x -k "stft or feature or label or separ or golden or e2e or wav or batches or eval" 2>&1 | tail -2
cd /tmp; timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_stft -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-graph --no-cpu-baseline > /dev/null 2>&1; cd $GRAFT_REPO_ROOT
f=$(find gpurun_out/prof_stft -name "*kernel_stats.csv" | head -1); grep -i "stft" $f | cut -c1-160
rm -rf gpurun_out/prof_stft
