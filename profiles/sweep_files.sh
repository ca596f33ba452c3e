#!/bin/bash
# A/B of whole source files on ONE box (boxes differ by ~5 %):  profiles/sweep_files.sh name=path/to/svgr_hip.hip ...
# each variant replaces csrc/svgr_hip.hip for its build; the tree's own file is restored at the end.  -> gpurun_out/sweep_files.txt
set -u
cd /tmp; export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
src=svgrasterize.py_amd/csrc/svgr_hip.hip
cp $src /tmp/svgr_hip.keep
out=gpurun_out/sweep_files.txt
: > $out
for v in "$@"; do
  name="${v%%=*}"; file="${v#*=}"
  cp "$file" $src
  make -s -C svgrasterize.py_amd/csrc clean
  if ! make -s -C svgrasterize.py_amd/csrc 2> gpurun_out/sweep_build_$name.err; then echo "$name BUILD FAILED" >> $out; continue; fi
  rm -rf gpurun_out/sf_$name
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/sf_$name -o t -- python3 bench.py --no-cpu-baseline --steps 40 > gpurun_out/sf_$name.log 2>&1 || { echo "$name RUN FAILED" >> $out; continue; }
  echo "== $name" >> $out
  python3 - $(find gpurun_out/sf_$name -name "*kernel_stats.csv" | head -1) >> $out <<'P'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if r["Name"].startswith(("k_", "void k_")) and int(r["Calls"]) > 5: print(f'  {r["Name"].split("(")[0][:40]:40s} {float(r["AverageNs"])/1e3:8.1f} us x {r["Calls"]}')
P
  tail -1 gpurun_out/sf_$name.log | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('   step', d['ms_per_step'], 'parity bad', (d.get('parity') or {}).get('bad'))" >> $out
  rm -rf gpurun_out/sf_$name
done
cp /tmp/svgr_hip.keep $src
make -s -C svgrasterize.py_amd/csrc clean && make -s -C svgrasterize.py_amd/csrc
cat $out
