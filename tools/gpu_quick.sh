#!/bin/bash
export TMPDIR=/tmp
timeout 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline 2> gpurun_out/b.err | python -c "
import json,sys
r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(r['ms_per_step'], r['value'], r.get('separate_dc_with_device_kmeans'))"
tail -2 gpurun_out/b.err
