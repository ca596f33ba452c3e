#!/bin/bash
# per-kernel durations of frames with new geometry:  profiles/replan_trace.sh [replan|cold]  -> gpurun_out/replan_trace_<mode>.txt
set -u
mode="${1:-replan}"
cd /tmp; export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
python3 profiles/scratch/replan_loop.py $mode 40 > gpurun_out/replan_trace_$mode.txt 2>&1
rm -rf gpurun_out/rt
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/rt -o t -- python3 profiles/scratch/replan_loop.py $mode 40 > gpurun_out/rt.log 2>&1 || { tail -5 gpurun_out/rt.log; exit 1; }
python3 - $(find gpurun_out/rt -name "*kernel_stats.csv" | head -1) >> gpurun_out/replan_trace_$mode.txt <<'P'
import csv, sys
tot = 0.0
for r in csv.DictReader(open(sys.argv[1])):
    print(f'  {r["Name"].split("(")[0][:60]:60s} {float(r["AverageNs"])/1e3:8.1f} us x {r["Calls"]}')
P
rm -rf gpurun_out/rt
cat gpurun_out/replan_trace_$mode.txt
