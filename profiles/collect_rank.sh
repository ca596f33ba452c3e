#!/bin/bash
# Counters of ONE rank of an N-way band-sharded render, on one GPU (profiles/emulate_rank.py: the same kernels and grids that
# rank runs under `bench.py --gpus N`):   profiles/collect_rank.sh <workload> <world> [rank]
#   -> gpurun_out/rank_<workload>_w<world>/pmc_kernels.json  (copy to profiles/pmc_kernels_<workload>_w<world>.json: bench.py's
#      roofline block of the N > 1 line reads it)
set -u
wl="${1:-synth8192}"; world="${2:-8}"; rank="${3:-0}"
cd /tmp; export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
o=gpurun_out/rank_${wl}_w${world}; rm -rf $o; mkdir -p $o
cmd="python3 profiles/emulate_rank.py --world $world --rank $rank --workload $wl --steps 12"
rocprofv3 --kernel-trace --stats --output-format csv -d $o/kt -o kt -- $cmd > $o/kt.log 2>&1 || exit 1
cp $(find $o/kt -name "*kernel_stats.csv" | head -1) $o/kernel_stats.csv
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAVES GRBM_GUI_ACTIVE --output-format csv -d $o/ps -o ps -- $cmd > $o/ps.log 2>&1 || exit 1
cp $(find $o/ps -name "*counter_collection.csv" | head -1) $o/pmc_sq.csv
cp $(find $o/ps -name "*kernel_trace.csv" | head -1) $o/pmc_sq_trace.csv
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $o/pf -o pf -- $cmd > $o/pf.log 2>&1 || exit 1
cp $(find $o/pf -name "*counter_collection.csv" | head -1) $o/pmc_fetch.csv
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $o/pw -o pw -- $cmd > $o/pw.log 2>&1 || exit 1
cp $(find $o/pw -name "*counter_collection.csv" | head -1) $o/pmc_write.csv
rm -rf $o/kt $o/pf $o/pw $o/ps
python3 profiles/pmc_json.py ${wl}_w${world} $o > $o/pmc_kernels.json
rm -f $o/pmc_sq.csv $o/pmc_sq_trace.csv $o/pmc_fetch.csv $o/pmc_write.csv
