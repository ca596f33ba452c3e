#!/bin/bash
# per-kernel average durations of build variants (kernel trace of a short bench run):
#   profiles/sweep_trace.sh "name:-DSVGR_..." ...   -> gpurun_out/sweep_trace.txt
set -u
cd /tmp; export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
out=gpurun_out/sweep_trace.txt
: > $out
for v in "$@"; do
  name="${v%%:*}"; flags="${v#*:}"
  make -s -C svgrasterize.py_amd/csrc clean
  if ! make -s -C svgrasterize.py_amd/csrc EXTRA="$flags" 2> gpurun_out/sweep_build_$name.err; then echo "$name BUILD FAILED" >> $out; continue; fi
  rm -rf gpurun_out/st_$name
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/st_$name -o t -- python3 bench.py --no-cpu-baseline --steps 40 > gpurun_out/st_$name.log 2>&1 || { echo "$name RUN FAILED" >> $out; continue; }
  echo "== $name" >> $out
  python3 - $(find gpurun_out/st_$name -name "*kernel_stats.csv" | head -1) >> $out <<'P'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if r["Name"].startswith(("k_", "void k_")): print(f'  {r["Name"].split("(")[0][:40]:40s} {float(r["AverageNs"])/1e3:8.1f} us x {r["Calls"]}')
P
  rm -rf gpurun_out/st_$name
done
make -s -C svgrasterize.py_amd/csrc clean && make -s -C svgrasterize.py_amd/csrc
cat $out
