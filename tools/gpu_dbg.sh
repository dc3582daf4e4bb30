#!/bin/bash
export TMPDIR=/tmp
ONSSEN_CHECK=1 timeout 200 python tools/cluster_probe.py 2>&1 | tail -8
