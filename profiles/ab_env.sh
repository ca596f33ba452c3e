#!/bin/bash
# A/B of run-time switches on ONE box and ONE build:  profiles/ab_env.sh "name:VAR=value VAR2=value" ...
#   per variant a kernel trace of a short bench run (per-kernel average us) -> gpurun_out/ab_env.txt; T_name also runs the GPU tests
set -u
cd /tmp; export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
out=gpurun_out/ab_env.txt
: > $out
for v in "$@"; do
  name="${v%%:*}"; envs="${v#*:}"
  echo "== $name  ($envs)" >> $out
  if [[ "$name" == T_* ]]; then
    env $envs timeout -k 10 600 python -m pytest tests -m gpu -x -q > gpurun_out/ab_pytest_$name.log 2>&1; echo "   pytest: $(tail -1 gpurun_out/ab_pytest_$name.log)" >> $out
    env $envs timeout -k 10 300 python3 bench.py --steps 40 2> gpurun_out/ab_bench_$name.err | tail -1 > gpurun_out/ab_bench_$name.json
    python3 -c "import json,sys; d=json.loads(open('gpurun_out/ab_bench_$name.json').read()); r=d['roofline']; print('   bench: step', d['ms_per_step'], 'tile', r['avg_launch_ms'], 'geo', r['geometry_ms'], 'parity', d.get('parity'))" >> $out 2>&1
  fi
  rm -rf gpurun_out/ab_$name
  ( export $envs; timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/ab_$name -o t -- python3 bench.py --no-cpu-baseline --steps 40 > gpurun_out/ab_$name.log 2>&1 ) || { echo "$name RUN FAILED" >> $out; tail -3 gpurun_out/ab_$name.log >> $out; continue; }
  python3 - $(find gpurun_out/ab_$name -name "*kernel_stats.csv" | head -1) >> $out <<'P'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if r["Name"].startswith(("k_", "void k_")) and int(r["Calls"]) > 5]
print("  " + "  ".join(f'{r["Name"].split("(")[0].replace("void ", "").split("<")[0][2:]} {float(r["AverageNs"])/1e3:.1f}' for r in rows),
      " | sum %.1f us" % (sum(float(r["AverageNs"]) for r in rows) / 1e3))
P
  rm -rf gpurun_out/ab_$name
done
cat $out
