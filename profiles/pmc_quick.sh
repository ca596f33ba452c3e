#!/bin/bash
# SQ counters of one kernel (PMC_KERNEL, default k_tile_render; a substring of the name) for run-time variants of ONE build:
#   [PMC_KERNEL="k_path_build<true>"] profiles/pmc_quick.sh "name:VAR=val" ...  -> gpurun_out/pmc_quick.txt
set -u
cd /tmp; export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
out=gpurun_out/pmc_quick.txt
: > $out
for v in "$@"; do
  name="${v%%:*}"; envs="${v#*:}"
  for pass in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAVES" "SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_INT32 SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SMEM SQ_INSTS_BRANCH"; do
    rm -rf gpurun_out/pq
    ( export $envs; timeout -k 10 300 rocprofv3 --kernel-trace --pmc $pass --output-format csv -d gpurun_out/pq -o pq -- python3 bench.py --no-cpu-baseline --steps 6 --warmup 2 > gpurun_out/pq.log 2>&1 ) || { echo "$name FAILED" >> $out; continue; }
    python3 - "$name" "${PMC_KERNEL:-k_tile_render}" $(find gpurun_out/pq -name "*counter_collection.csv" | head -1) >> $out <<'P'
import csv, sys, collections
acc = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[3])):
    if sys.argv[2] in r["Kernel_Name"]:
        acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
# (one row per dispatch and counter dimension instance: sum the instances of a dispatch = total / dispatches)
print(sys.argv[1], {k: round(sum(v) / len(v) / 1e6, 3) for k, v in acc.items()}, "(millions per launch)")
P
  done
done
rm -rf gpurun_out/pq
cat $out
