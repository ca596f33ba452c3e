#!/bin/bash
# Collect the judged profile artifacts on the GPU box: profiles/collect2.sh <tag> [workload]     (e.g. r02_c2 synth4096)
#   gpurun_out/<tag>/bench.json          bench line (N=1)
#   gpurun_out/<tag>/kernel_stats.csv    rocprofv3 --kernel-trace --stats of the same command
#   gpurun_out/<tag>/pmc_sq.csv          --pmc pass 1: SQ instruction / cycle counters + GRBM_GUI_ACTIVE (clock)
#   gpurun_out/<tag>/pmc_fetch.csv, pmc_write.csv   --pmc passes 2, 3: FETCH_SIZE, WRITE_SIZE (they do not fit one pass)
#   gpurun_out/<tag>/pmc_mix.csv         --pmc pass 4: the f64 / int32 split of the VALU instructions
#   gpurun_out/<tag>/pmc_kernels.json    per kernel: average counters per launch (profiles/pmc_json.py); bench.py reads the
#                                        copy committed as profiles/pmc_kernels_<workload>.json for its roofline block
# The profiled runs carry --no-configs: the counters describe THIS workload alone (the default bench line also draws the other BASELINE
# configurations in the same process).  Counter passes carry --kernel-trace only (no other trace domain).  Copy what should be judged into profiles/<round>/.
set -u
tag="${1:-run}"; wl="${2:-synth4096}"
cd /tmp; export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
o=gpurun_out/$tag; rm -rf $o; mkdir -p $o
rocprofv3 --kernel-trace --stats --output-format csv -d $o/kt -o kt -- python3 bench.py --workload $wl --no-cpu-baseline --no-configs --steps ${KT_STEPS:-50} > $o/kt.log 2>&1 || exit 1
cp $(find $o/kt -name "*kernel_stats.csv" | head -1) $o/kernel_stats.csv
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAVES GRBM_GUI_ACTIVE --output-format csv -d $o/ps -o ps -- python3 bench.py --workload $wl --no-cpu-baseline --no-configs --steps 10 --warmup 2 > $o/ps.log 2>&1 || exit 1
cp $(find $o/ps -name "*counter_collection.csv" | head -1) $o/pmc_sq.csv
cp $(find $o/ps -name "*kernel_trace.csv" | head -1) $o/pmc_sq_trace.csv
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $o/pf -o pf -- python3 bench.py --workload $wl --no-cpu-baseline --no-configs --steps 10 --warmup 2 > $o/pf.log 2>&1 || exit 1
cp $(find $o/pf -name "*counter_collection.csv" | head -1) $o/pmc_fetch.csv
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $o/pw -o pw -- python3 bench.py --workload $wl --no-cpu-baseline --no-configs --steps 10 --warmup 2 > $o/pw.log 2>&1 || exit 1
cp $(find $o/pw -name "*counter_collection.csv" | head -1) $o/pmc_write.csv
rocprofv3 --pmc SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_INT32 --output-format csv -d $o/pm -o pm -- python3 bench.py --workload $wl --no-cpu-baseline --no-configs --steps 10 --warmup 2 > $o/pm.log 2>&1 || exit 1
cp $(find $o/pm -name "*counter_collection.csv" | head -1) $o/pmc_mix.csv
rm -rf $o/kt $o/pf $o/pw $o/ps $o/pm
python3 profiles/pmc_json.py $wl $o > $o/pmc_kernels.json
# the bench line LAST, priced with the counters just collected (it used to run first and reported the previous build's)
cp $o/pmc_kernels.json profiles/pmc_kernels_$wl.json
python3 bench.py --workload $wl --steps ${STEPS:-200} > $o/bench.json 2> $o/bench.err || exit 1
# (scene workloads dispatch thousands of kernels: the raw per-dispatch tables would not fit the 64 MiB that travel back)
for f in $o/pmc_sq.csv $o/pmc_sq_trace.csv $o/pmc_fetch.csv $o/pmc_write.csv $o/pmc_mix.csv; do [ $(stat -c %s $f) -gt 4000000 ] && rm -f $f; done
cat $o/pmc_kernels.json
cut -c1-140 $o/kernel_stats.csv | head -10
cat $o/bench.json
