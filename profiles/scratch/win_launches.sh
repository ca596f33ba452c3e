#!/bin/bash
# grid size and duration of every window launch of the tile kernel in warm icons renders
set -u
cd /tmp; export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/wl
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/wl -o t -- python3 bench.py --workload ${1:-icons4096} --no-cpu-baseline --steps 4 --warmup 2 > gpurun_out/wl.log 2>&1 || { tail -5 gpurun_out/wl.log; exit 1; }
python3 - $(find gpurun_out/wl -name "*kernel_trace.csv" | head -1) <<'P'
import csv, sys, collections
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
sel = [r for r in rows if "k_tile_render<1, true, true, true>" in r["Kernel_Name"]][-30:]
tot = 0
for r in sel:
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    tot += d
    print(f'  WGs {int(r["Grid_Size_X"]) // int(r["Workgroup_Size_X"]):6d}  {d:7.1f} us  vgpr {r["VGPR_Count"]} lds {r["LDS_Block_Size"]}')
print("sum", round(tot, 1), "us over", len(sel), "launches")
P
rm -rf gpurun_out/wl
