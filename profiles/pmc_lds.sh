#!/bin/bash
# LDS-side counters of the kernels (diagnostic, two --pmc passes): how busy the LDS pipeline is next to the VALU.
#   profiles/pmc_lds.sh [workload]   ->  gpurun_out/pmc_lds.txt (per kernel averages)
set -u
wl="${1:-synth4096}"
cd /tmp; export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
o=gpurun_out/pmc_lds; rm -rf $o; mkdir -p $o
rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_LDS_ATOMIC GRBM_GUI_ACTIVE --output-format csv -d $o/a -o a -- python3 bench.py --workload $wl --no-cpu-baseline --steps 10 --warmup 2 > $o/a.log 2>&1 || exit 1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_INT32 SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $o/b -o b -- python3 bench.py --workload $wl --no-cpu-baseline --steps 10 --warmup 2 > $o/b.log 2>&1 || exit 1
python3 - $o <<'P' > gpurun_out/pmc_lds.txt
import csv, glob, sys, collections
o = sys.argv[1]
for tag in ("a", "b"):
    f = glob.glob(f"{o}/{tag}/**/*counter_collection.csv", recursive=True)[0]
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"].split("(")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in acc.items():
        if not k.startswith(("k_", "void k_")): continue
        print(k, {c: round(sum(x) / len(x), 1) for c, x in v.items()})
P
rm -rf $o
cat gpurun_out/pmc_lds.txt
