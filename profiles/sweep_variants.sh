#!/bin/bash
# Build tile-geometry variants of the library on the GPU box and bench each (diagnostic; results in gpurun_out/sweep.txt).
# usage: profiles/sweep_variants.sh "name1:-DSVGR_X=.. -DSVGR_Y=.." "name2:..."
set -u
cd "$GRAFT_REPO_ROOT"
out=gpurun_out/sweep.txt
: > $out
for v in "$@"; do
  name="${v%%:*}"; flags="${v#*:}"
  make -s -C svgrasterize.py_amd/csrc clean
  if ! make -s -C svgrasterize.py_amd/csrc EXTRA="$flags" 2> gpurun_out/sweep_build_$name.err; then echo "$name BUILD FAILED" >> $out; continue; fi
  timeout -k 10 120 python bench.py --no-cpu-baseline --steps 30 > gpurun_out/sweep_$name.json 2> gpurun_out/sweep_$name.err || { echo "$name RUN FAILED" >> $out; continue; }
  python - "$name" gpurun_out/sweep_$name.json >> $out <<'P'
import json,sys
d=json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
print(sys.argv[1], "step", d["ms_per_step"], "tile", d["roofline"]["avg_launch_ms"], "geo", d["roofline"]["geometry_ms"])
P
done
make -s -C svgrasterize.py_amd/csrc clean && make -s -C svgrasterize.py_amd/csrc
cat $out
