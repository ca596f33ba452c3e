#!/bin/bash
# A/B of an environment switch on one box: ab_env.sh VAR   (runs bench under kernel trace with VAR unset / =1, twice)
cd /tmp; export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
out=gpurun_out/ab_env.txt; : > $out
for i in 1 2; do for v in off on; do
  rm -rf gpurun_out/ab_$v
  if [ $v = on ]; then export $1=1; else unset $1; fi
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/ab_$v -o t -- python3 bench.py --no-cpu-baseline --steps 40 > gpurun_out/ab_$v.log 2>&1
  echo "== $1 $v" >> $out
  python3 - $(find gpurun_out/ab_$v -name "*kernel_stats.csv" | head -1) >> $out <<'P'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if r["Name"].startswith(("k_", "void k_")) and int(r["Calls"]) > 5: print(f'  {r["Name"].split("(")[0][:40]:40s} {float(r["AverageNs"])/1e3:8.1f} us x {r["Calls"]}')
P
  tail -1 gpurun_out/ab_$v.log | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('   step', d['ms_per_step'])" >> $out
  rm -rf gpurun_out/ab_$v
done; done
cat $out
