#!/bin/bash
# round 6, call m: Lloyd kernel with c1 - c0 in scalar registers against the LDS re-reads, same box
export TMPDIR=/tmp
mkdir -p gpurun_out
python tools/ab_bits.py base kmsgpr 2>&1 | tail -2
python - <<'P'
import json, os, subprocess, sys
for rep in range(2):
    for v in ("base", "kmsgpr"):
        env = dict(os.environ, ONSSEN_HIP_LIB=os.path.abspath(f"build_variants/libonssen_hip_{v}.so"))
        out = subprocess.run([sys.executable, "bench.py", "--no-cpu-baseline", "--no-extra", "--steps", "40"], env=env, capture_output=True, text=True).stdout.strip().splitlines()
        r = json.loads(out[-1])
        legs = r["roofline"].get("dc_back_end_legs_ms", {})
        print(v, "ms_per_step %.4f" % r["ms_per_step"], "cluster leg %.4f" % r["roofline"]["legs_ms"]["threshold_2means_masks"], {k: round(x, 4) for k, x in legs.items()} if isinstance(legs, dict) else legs, "lloyd mean", r["lloyd_iterations"]["mean"], flush=True)
P
ONSSEN_HIP_LIB=$PWD/build_variants/libonssen_hip_kmsgpr.so timeout 900 python -m pytest tests -m gpu -q -x -k "cluster or kmeans or dc_masks or separate or smoke or trained or ragged or c_abi" 2>&1 | tail -3
