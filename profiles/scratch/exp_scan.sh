#!/bin/bash
# A/B of k_flatten<.., SCAN> build variants on one box: profiles/scratch/exp_scan.sh "name:-D..." ...
cd /tmp; export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
for v in "$@"; do
  name="${v%%:*}"; flags="${v#*:}"
  make -s -C svgrasterize.py_amd/csrc clean; make -s -C svgrasterize.py_amd/csrc EXTRA="$flags" > /dev/null 2>&1 || { echo "$name BUILD FAILED"; continue; }
  rm -rf gpurun_out/es
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/es -o t -- python3 profiles/scratch/replan_loop.py replan 40 > gpurun_out/es.log 2>&1
  python3 - "$name" $(find gpurun_out/es -name "*kernel_stats.csv" | head -1) <<'P'
import csv, sys
for r in csv.DictReader(open(sys.argv[2])):
    if "k_flatten<true, false" in r["Name"]: print(sys.argv[1], "SCAN flatten %.1f us x %s" % (float(r["AverageNs"]) / 1e3, r["Calls"]))
P
  tail -1 gpurun_out/es.log
done
make -s -C svgrasterize.py_amd/csrc clean; make -s -C svgrasterize.py_amd/csrc > /dev/null 2>&1
