#!/bin/bash
set -u
cd /tmp; export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/wl
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/wl -o t -- python3 bench.py --workload icons4096 --no-cpu-baseline --steps 4 --warmup 2 > gpurun_out/wl.log 2>&1 || { tail -5 gpurun_out/wl.log; exit 1; }
python3 - $(find gpurun_out/wl -name "*kernel_trace.csv" | head -1) <<'P'
import csv, sys
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
sel = [i for i, r in enumerate(rows) if "k_tile_render<1, true, true, true>" in r["Kernel_Name"]][-30:]
t0 = int(rows[sel[0] - 8]["Start_Timestamp"])
for i in range(sel[0] - 8, sel[-1] + 4):
    r = rows[i]
    print(f'  {(int(r["Start_Timestamp"]) - t0) / 1e3:8.1f} -> {(int(r["End_Timestamp"]) - t0) / 1e3:8.1f} us  q{r["Queue_Id"]} s{r["Stream_Id"]}  {r["Kernel_Name"].split("(")[0][:44]}')
P
rm -rf gpurun_out/wl
