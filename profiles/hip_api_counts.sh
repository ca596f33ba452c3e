#!/bin/bash
# HIP API calls of a scene workload's warm steps (hip trace only, no counters):  profiles/hip_api_counts.sh [workload] [steps] -> gpurun_out/hip_api.txt
set -u
wl="${1:-icons4096}"; steps="${2:-10}"
cd /tmp; export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/ha
timeout -k 10 300 rocprofv3 --hip-trace --kernel-trace --stats --output-format csv -d gpurun_out/ha -o t -- python3 bench.py --workload $wl --no-cpu-baseline --steps $steps --warmup 2 > gpurun_out/ha.log 2>&1 || { tail -5 gpurun_out/ha.log; exit 1; }
{
echo "== $wl, $steps steps + 2 warm-up + the cold passes of the bench line"
for f in hip_api_stats kernel_stats; do
  echo "-- $f"; python3 - $(find gpurun_out/ha -name "*${f}.csv" | head -1) <<'P'
import csv, sys
for r in list(csv.DictReader(open(sys.argv[1])))[:22]:
    print(f'  {r["Name"].split("(")[0][:48]:48s} calls {r["Calls"]:>7s}  avg {float(r["AverageNs"])/1e3:8.1f} us  total {float(r["TotalDurationNs"])/1e6:8.2f} ms')
P
done
tail -1 gpurun_out/ha.log | cut -c1-300
} > gpurun_out/hip_api.txt
rm -rf gpurun_out/ha
cat gpurun_out/hip_api.txt
