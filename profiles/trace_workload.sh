#!/bin/bash
# per-kernel average durations of one bench workload (kernel trace):  profiles/trace_workload.sh tiger2048 [bench args]  -> gpurun_out/trace_<workload>.txt
set -u
wl="${1:-tiger2048}"; shift || true
cd /tmp; export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/tw
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/tw -o t -- python3 bench.py --workload $wl --no-cpu-baseline --steps 100 "$@" > gpurun_out/tw.log 2>&1 || { tail -5 gpurun_out/tw.log; exit 1; }
python3 - $(find gpurun_out/tw -name "*kernel_stats.csv" | head -1) > gpurun_out/trace_$wl.txt <<'P'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if r["Name"].startswith(("k_", "void k_")): print(f'  {r["Name"].split("(")[0][:44]:44s} {float(r["AverageNs"])/1e3:8.1f} us x {r["Calls"]}')
P
tail -1 gpurun_out/tw.log | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('  step', d['ms_per_step'], 'tile', d['roofline']['avg_launch_ms'], 'geometry', d['roofline']['geometry_ms'], 'P', d['config']['path_pixels'], 'edges', d['config']['edges'])" >> gpurun_out/trace_$wl.txt
rm -rf gpurun_out/tw
cat gpurun_out/trace_$wl.txt
