#!/bin/bash
# profiles/scratch/emu_variant.sh "name:-Dflags" ... : per variant the 1-GPU kernel trace and the 1- / 8-rank emulation of config 4
set -u
cd /tmp; export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
out=gpurun_out/emu_variant.txt; : > $out
for v in "$@"; do
  name="${v%%:*}"; flags="${v#*:}"
  make -s -C svgrasterize.py_amd/csrc clean; make -s -C svgrasterize.py_amd/csrc EXTRA="$flags" 2> /dev/null || { echo "$name BUILD FAILED" >> $out; continue; }
  echo "== $name ($flags)" >> $out
  rm -rf gpurun_out/ev; timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/ev -o t -- python3 bench.py --no-cpu-baseline --steps 40 > gpurun_out/ev.log 2>&1
  python3 - $(find gpurun_out/ev -name "*kernel_stats.csv" | head -1) >> $out <<'P'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if r["Name"].startswith(("k_", "void k_")) and int(r["Calls"]) > 20]
print("  1 GPU synth4096: " + "  ".join(f'{r["Name"].split("(")[0].replace("void ", "").split("<")[0][2:]} {float(r["AverageNs"])/1e3:.1f}' for r in rows))
P
  for w in 1 8; do python profiles/emulate_rank.py --world $w --all --workload synth8192 --steps 40 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('  emulate', json.dumps(dict(world=d['world'], slowest_ms=d['slowest']['ms_step'], ranks_ms=[r['ms_step'] for r in d['ranks']])))" >> $out; done
  rm -rf gpurun_out/ev
done
make -s -C svgrasterize.py_amd/csrc clean && make -s -C svgrasterize.py_amd/csrc
cat $out
