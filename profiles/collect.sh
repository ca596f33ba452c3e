#!/bin/bash
# Collect the judged profile artifacts on the GPU box: profiles/collect.sh <tag>   (e.g. r01_v4)
#   gpurun_out/<tag>/bench.json               bench line (N=1, default workload)
#   gpurun_out/<tag>/kernel_stats.csv         rocprofv3 --kernel-trace --stats of the same command
#   gpurun_out/<tag>/pmc_fetch.csv, pmc_write.csv   separate --pmc passes (FETCH_SIZE / WRITE_SIZE)
# Copy what should be judged into profiles/<round>/ afterwards (gpurun_out/ is scratch).
set -u
tag="${1:-run}"
cd /tmp; export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
o=gpurun_out/$tag; rm -rf $o; mkdir -p $o
python3 bench.py > $o/bench.json 2> $o/bench.err || exit 1
rocprofv3 --kernel-trace --stats --output-format csv -d $o/kt -o kt -- python3 bench.py --no-cpu-baseline --steps 50 > $o/kt.log 2>&1 || exit 1
cp $(find $o/kt -name "*kernel_stats.csv" | head -1) $o/kernel_stats.csv
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $o/pf -o pf -- python3 bench.py --no-cpu-baseline --steps 10 --warmup 2 > $o/pf.log 2>&1 || exit 1
cp $(find $o/pf -name "*counter_collection.csv" | head -1) $o/pmc_fetch.csv
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $o/pw -o pw -- python3 bench.py --no-cpu-baseline --steps 10 --warmup 2 > $o/pw.log 2>&1 || exit 1
cp $(find $o/pw -name "*counter_collection.csv" | head -1) $o/pmc_write.csv
rm -rf $o/kt $o/pf $o/pw
python3 profiles/pmc_summary.py --only=k_tile_render $o/pmc_fetch.csv $o/pmc_write.csv
head -8 $o/kernel_stats.csv | cut -c1-120
cat $o/bench.json
