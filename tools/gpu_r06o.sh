#!/bin/bash
# round 6, call o: kernel trace of the pipelined headline step
export TMPDIR=/tmp
mkdir -p gpurun_out
python tools/micro/pipe2_profile.py 40
cd /tmp && PIPE2_GRAPH=0 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_p2 -- python $GRAFT_REPO_ROOT/tools/micro/pipe2_profile.py 40 > /tmp/prof_p2.log 2>&1
cd $GRAFT_REPO_ROOT
tail -2 /tmp/prof_p2.log
f=$(find /tmp/prof_p2 -name "*kernel_stats.csv" | head -1)
cp "$f" gpurun_out/r06o_pipe2_kernel_stats.csv
python - <<'P'
import csv
for r in list(csv.DictReader(open("gpurun_out/r06o_pipe2_kernel_stats.csv")))[:16]:
    print(f"{r['Name'][:100]:100s} {r['Calls']:>5s} {float(r['AverageNs'])/1e3:9.1f} us  {r['Percentage']}")
P
