#!/bin/bash
export TMPDIR=/tmp
for s in 0 64 128 32 0 64; do
  ONSSEN_X3Q_SPLIT=$s timeout 300 python bench.py --no-extra --no-cpu-baseline --steps 60 --warmup 5 > gpurun_out/ab_s$s.json 2>/dev/null
  python - <<PY
import json
r = json.loads(open("gpurun_out/ab_s$s.json").read().strip().splitlines()[-1])
print("split=$s headline ms/step %.4f resident %.4f" % (r["ms_per_step"], r["resident_mask_step"]["ms_per_step"]), {k: round(v, 4) for k, v in r["roofline"]["other_kernels"]["ms_by_call"].items()})
PY
done
for s in 0 64; do
  ONSSEN_X3Q_SPLIT=$s timeout 300 python bench.py --config chimera_l4 --no-extra --no-cpu-baseline --steps 30 --warmup 3 2>/dev/null | python -c "
import json,sys; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('chimera split=$s ms/step %.4f' % r['ms_per_step'], {k: round(v, 4) for k, v in r['roofline']['other_kernels']['ms_by_call'].items()})"
done
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "linear or golden or cfg3 or chimera" 2>&1 | tail -2
