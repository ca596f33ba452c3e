#!/bin/bash
# A/B of ENVIRONMENT switches on the built library, one box:  profiles/ab_trace.sh [--pytest] "name:VAR=1 VAR2=x" ...
#   per variant: kernel trace of a short bench run (per-kernel average us) and the bench line's step / parity -> gpurun_out/ab_trace.txt
set -u
cd /tmp; export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
out=gpurun_out/ab_trace.txt
: > $out
if [ "${1:-}" == "--pytest" ]; then
  shift
  timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/ab_pytest.log 2>&1; echo "pytest: $(tail -1 gpurun_out/ab_pytest.log)" >> $out
fi
for v in "$@"; do
  name="${v%%:*}"; envs="${v#*:}"
  echo "== $name  ($envs)" >> $out
  ( [ -n "$envs" ] && export $envs; timeout -k 10 300 python3 bench.py --steps 40 2> gpurun_out/ab_bench_$name.err | tail -1 > gpurun_out/ab_bench_$name.json )
  python3 -c "import json,sys; d=json.loads(open('gpurun_out/ab_bench_$name.json').read()); r=d['roofline']; print('   bench: step', d['ms_per_step'], 'tile', r['avg_launch_ms'], 'geo', r['geometry_ms'], 'cold', d.get('cold_ms'), 'replan', d.get('replan_ms'), 'plan', d.get('plan_ms'), 'parity', d.get('parity'))" >> $out 2>&1
  rm -rf gpurun_out/ab_$name
  ( [ -n "$envs" ] && export $envs; timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/ab_$name -o t -- python3 bench.py --no-cpu-baseline --steps 40 > gpurun_out/ab_$name.log 2>&1 ) || { echo "$name RUN FAILED" >> $out; tail -3 gpurun_out/ab_$name.log >> $out; continue; }
  python3 - $(find gpurun_out/ab_$name -name "*kernel_stats.csv" | head -1) >> $out <<'P'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if r["Name"].startswith(("k_", "void k_")) and int(r["Calls"]) > 5]
print("  " + "  ".join(f'{r["Name"].split("(")[0].replace("void ", "").split("<")[0][2:]} {float(r["AverageNs"])/1e3:.1f}' for r in rows),
      " | sum %.1f us" % (sum(float(r["AverageNs"]) for r in rows) / 1e3))
P
  rm -rf gpurun_out/ab_$name
done
cat $out
