#!/bin/bash
export TMPDIR=/tmp
timeout 200 python tools/train_soak.py --steps 300 2>&1 < /dev/null | tail -3 | cut -c1-600
