#!/bin/bash
# Scratch wrapper for one gpurun call while iterating (edit freely).
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "stft or istft or e2e or separ or smoke or cfg5 or loader" 2>&1 | tail -3
timeout 300 python bench.py --no-cpu-baseline --no-extra > gpurun_out/bench_fft.json 2>/dev/null; python - <<'PY'
import json
r=json.loads(open('gpurun_out/bench_fft.json').read().strip().splitlines()[-1])
print("headline", r["ms_per_step"], r["roofline"]["hbm_kernels"], r["roofline"]["legs_sum_ms"])
PY
