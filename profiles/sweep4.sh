#!/bin/bash
# Round-4 A/B of build variants on ONE box:  profiles/sweep4.sh "name:-DSVGR_X_...=1 ..." ...
#   per variant: kernel trace of a short bench run (per-kernel average us) -> gpurun_out/sweep4.txt
#   a variant whose name starts with T_ also runs the GPU test suite and a bench line with the whole-canvas parity check
set -u
cd /tmp; export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
out=gpurun_out/sweep4.txt
: > $out
for v in "$@"; do
  name="${v%%:*}"; flags="${v#*:}"; envs=""
  if [[ "$flags" == *";"* ]]; then envs="${flags%%;*}"; flags="${flags#*;}"; fi
  make -s -C svgrasterize.py_amd/csrc clean
  if ! make -s -C svgrasterize.py_amd/csrc EXTRA="$flags" 2> gpurun_out/sweep4_build_$name.err; then echo "$name BUILD FAILED" >> $out; continue; fi
  rm -rf gpurun_out/s4_$name
  echo "== $name  ($flags)" >> $out
  if [[ "$name" == T_* ]]; then
    env $envs timeout -k 10 600 python -m pytest tests -m gpu -x -q > gpurun_out/sweep4_pytest_$name.log 2>&1; echo "   pytest: $(tail -1 gpurun_out/sweep4_pytest_$name.log)" >> $out
    env $envs timeout -k 10 300 python3 bench.py --steps 40 2> gpurun_out/sweep4_bench_$name.err | tail -1 > gpurun_out/sweep4_bench_$name.json
    python3 -c "import json,sys; d=json.loads(open('gpurun_out/sweep4_bench_$name.json').read()); r=d['roofline']; print('   bench: step', d['ms_per_step'], 'tile', r['avg_launch_ms'], 'geo', r['geometry_ms'], 'parity', d.get('parity'))" >> $out 2>&1
  fi
  ( [ -n "$envs" ] && export $envs; timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/s4_$name -o t -- python3 bench.py --no-cpu-baseline --steps 40 > gpurun_out/s4_$name.log 2>&1 ) || { echo "$name RUN FAILED" >> $out; tail -3 gpurun_out/s4_$name.log >> $out; continue; }
  python3 - $(find gpurun_out/s4_$name -name "*kernel_stats.csv" | head -1) >> $out <<'P'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if r["Name"].startswith(("k_", "void k_")) and int(r["Calls"]) > 5]
print("  " + "  ".join(f'{r["Name"].split("(")[0].replace("void ", "").split("<")[0][2:]} {float(r["AverageNs"])/1e3:.1f}' for r in rows),
      " | sum %.1f us" % (sum(float(r["AverageNs"]) for r in rows) / 1e3))
P
  rm -rf gpurun_out/s4_$name
done
make -s -C svgrasterize.py_amd/csrc clean && make -s -C svgrasterize.py_amd/csrc
cat $out
