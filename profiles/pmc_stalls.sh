#!/bin/bash
# Where a wave of the tile kernel spends its cycles (diagnostic; separate --pmc passes):  profiles/pmc_stalls.sh  -> gpurun_out/pmc_stalls.txt
set -u
cd /tmp; export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
out=gpurun_out/pmc_stalls.txt
: > $out
for pass in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS" \
            "SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_FLAT SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_IFETCH SQ_INSTS_VALU SQ_BUSY_CU_CYCLES" \
            "SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ATOMIC_RETURN SQ_INSTS_LDS SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_LEVEL_WAVES" \
            "SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_INSTS_SENDMSG SQ_INSTS_VSKIPPED SQ_WAIT_INST_ANY"; do
  rm -rf gpurun_out/pq
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $pass --output-format csv -d gpurun_out/pq -o pq -- python3 bench.py --no-cpu-baseline --steps 6 --warmup 2 > gpurun_out/pq.log 2>&1 || { echo "pass FAILED: $pass" >> $out; tail -3 gpurun_out/pq.log >> $out; continue; }
  python3 - $(find gpurun_out/pq -name "*counter_collection.csv" | head -1) >> $out <<'P'
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"].split("(")[0].replace("void ", "")
    if k.startswith("k_tile_render") or k.startswith("k_path_build"):
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in acc.items():
    print(k, {c: round(sum(x) / len(x) / 1e6, 3) for c, x in v.items()}, "(millions per launch)")
P
done
rm -rf gpurun_out/pq
cat $out
