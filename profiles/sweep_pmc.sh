#!/bin/bash
# SQ instruction counters of one kernel across build variants:  profiles/sweep_pmc.sh <kernel-prefix> "name:-DSVGR_..." ...
# -> gpurun_out/sweep_pmc.txt   (counter passes carry --kernel-trace only)
set -u
cd /tmp; export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
kern="$1"; shift
out=gpurun_out/sweep_pmc.txt
: > $out
for v in "$@"; do
  name="${v%%:*}"; flags="${v#*:}"
  make -s -C svgrasterize.py_amd/csrc clean
  if ! make -s -C svgrasterize.py_amd/csrc EXTRA="$flags" 2> gpurun_out/sweep_build_$name.err; then echo "$name BUILD FAILED" >> $out; continue; fi
  rm -rf gpurun_out/sp_$name
  rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES --output-format csv -d gpurun_out/sp_$name -o t -- python3 bench.py --no-cpu-baseline --steps 8 --warmup 2 > gpurun_out/sp_$name.log 2>&1 || { echo "$name RUN FAILED" >> $out; continue; }
  echo "== $name" >> $out
  python3 - "$kern" $(find gpurun_out/sp_$name -name "*counter_collection.csv" | head -1) >> $out <<'P'
import csv, sys, collections
acc = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[2])):
    if r["Kernel_Name"].startswith(sys.argv[1]) or r["Kernel_Name"].startswith("void " + sys.argv[1]):
        acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
print("  " + "  ".join(f"{k} {sum(v)/len(v)/1e6:.2f}M" for k, v in sorted(acc.items())))
P
  rm -rf gpurun_out/sp_$name
done
make -s -C svgrasterize.py_amd/csrc clean && make -s -C svgrasterize.py_amd/csrc
cat $out
