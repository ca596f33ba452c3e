#!/bin/bash
# Like sweep_variants.sh for geometry-side build variants: per-kernel averages of the synthetic bench + tiger / material wall clocks.
set -u
cd /tmp; export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
out=gpurun_out/sweepg.txt
: > $out
for v in "$@"; do
  name="${v%%:*}"; flags="${v#*:}"
  make -s -C svgrasterize.py_amd/csrc clean
  if ! make -s -C svgrasterize.py_amd/csrc EXTRA="$flags" 2> gpurun_out/sweepg_build_$name.err; then echo "$name BUILD FAILED" >> $out; continue; fi
  rm -rf gpurun_out/sweepg_$name
  timeout -k 10 200 rocprofv3 --kernel-trace -d gpurun_out/sweepg_$name -o t -- python3 bench.py --no-cpu-baseline --steps 30 > gpurun_out/sweepg_$name.log 2>&1 || { echo "$name RUN FAILED" >> $out; }
  echo "== $name" >> $out
  python3 - gpurun_out/sweepg_$name/t_results.db >> $out <<'P'
import sqlite3, sys
c = sqlite3.connect(sys.argv[1])
for r in c.execute("select name, count(*), avg(end-start) from kernels group by name order by 3 desc").fetchall():
    if r[1] > 5: print("  ", r[0][:30].ljust(32), r[1], round(r[2]/1e3,2))
P
  tail -1 gpurun_out/sweepg_$name.log | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('   step', d['ms_per_step'])" >> $out
  timeout -k 10 200 python3 bench_scenes.py --repeat 3 2>/dev/null | python3 -c "
import json,sys
for l in sys.stdin:
    d=json.loads(l); print('   ', d['scene'][:28], d['render_s'])" >> $out
  rm -rf gpurun_out/sweepg_$name
done
make -s -C svgrasterize.py_amd/csrc clean && make -s -C svgrasterize.py_amd/csrc
cat $out
