#!/usr/bin/env python
"""Same-box A/B of an environment switch: python tools/ab_env.py VAR v1 v2 ... -- bench.py args...  (every value twice, interleaved)."""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
i = sys.argv.index("--")
var, vals, cmd = sys.argv[1], sys.argv[2:i], sys.argv[i + 1:]
for rep in range(2):
    for v in vals:
        env = dict(os.environ); env[var] = v
        out = subprocess.run([sys.executable] + cmd, cwd=ROOT, capture_output=True, text=True, timeout=900, env=env).stdout.strip().splitlines()
        try:
            r = json.loads(out[-1])
            gem = (r.get("roofline", {}).get("other_kernels") or {}).get("ms_by_call") or {}
            print(f"{var}={v:6s} ms_per_step {r['ms_per_step']:.4f}  one-batch {r.get('one_batch_at_a_time_step', {}).get('ms_per_step', 0):.4f}  gemm ms "
                  + " ".join(f"{k}={x:.4f}" for k, x in gem.items() if x), flush=True)
        except Exception:
            print(f"{var}={v} FAILED: {out[-3:]}", flush=True)
