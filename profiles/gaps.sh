#!/bin/bash
# idle time between the kernels of a steady-state step (kernel trace only, no counters):  profiles/gaps.sh [bench args]  -> gpurun_out/gaps.txt
set -u
cd /tmp; export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/gp
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/gp -o t -- python3 bench.py --no-cpu-baseline --steps 40 "$@" > gpurun_out/gp.log 2>&1 || { tail -5 gpurun_out/gp.log; exit 1; }
python3 - $(find gpurun_out/gp -name "*kernel_trace.csv" | head -1) > gpurun_out/gaps.txt <<'P'
import csv, sys, collections
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
# the last 30 steps: a step ends with the tile kernel
names = [r["Kernel_Name"].split("(")[0].replace("void ", "") for r in rows]
ends = [i for i, n in enumerate(names) if n.startswith("k_tile_render")]
first = ends[-31] + 1
gap = collections.defaultdict(list); dur = collections.defaultdict(list); steps = []
prev_end = int(rows[first - 1]["End_Timestamp"]); step_start = None
for i in range(first, ends[-1] + 1):
    s, e = int(rows[i]["Start_Timestamp"]), int(rows[i]["End_Timestamp"])
    gap[names[i]].append(s - prev_end); dur[names[i]].append(e - s)
    if step_start is None: step_start = prev_end
    if names[i].startswith("k_tile_render"):
        steps.append(e - step_start); step_start = None
    prev_end = e
print(f"steady state, {len(steps)} steps: end of a tile kernel to the end of the next = {sum(steps)/len(steps)/1e3:.1f} us")
for n in dur:
    print(f"  {n[:44]:44s} x{len(dur[n])/len(steps):.0f}  runs {sum(dur[n])/len(dur[n])/1e3:7.1f} us, starts {sum(gap[n])/len(gap[n])/1e3:6.2f} us after the kernel in front of it ended")
P
rm -rf gpurun_out/gp
cat gpurun_out/gaps.txt
