#!/bin/bash
# Like sweep_variants.sh, but reports per-kernel averages from a rocprofv3 kernel trace (diagnostic builds may render garbage).
set -u
cd /tmp; export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
out=gpurun_out/sweepk.txt
: > $out
for v in "$@"; do
  name="${v%%:*}"; flags="${v#*:}"
  make -s -C svgrasterize.py_amd/csrc clean
  if ! make -s -C svgrasterize.py_amd/csrc EXTRA="$flags" 2> gpurun_out/sweepk_build_$name.err; then echo "$name BUILD FAILED" >> $out; continue; fi
  rm -rf gpurun_out/sweepk_$name
  timeout -k 10 200 rocprofv3 --kernel-trace -d gpurun_out/sweepk_$name -o t -- python3 ${SWEEP_CMD:-bench.py --no-cpu-baseline --steps 30} > gpurun_out/sweepk_$name.log 2>&1 || { echo "$name RUN FAILED" >> $out; }
  echo "== $name" >> $out
  python3 - gpurun_out/sweepk_$name/t_results.db >> $out <<'P'
import sqlite3, sys
c = sqlite3.connect(sys.argv[1])
for r in c.execute("select name, count(*), avg(end-start) from kernels group by name order by 3 desc").fetchall():
    if r[1] > 5: print("  ", r[0][:30].ljust(32), r[1], round(r[2]/1e3,2))
P
done
make -s -C svgrasterize.py_amd/csrc clean && make -s -C svgrasterize.py_amd/csrc
cat $out
