#!/bin/bash
# copy what profiles/collect_round.sh <tag> left in gpurun_out/ into the tracked record:  profiles/publish_round.sh <tag> <round-dir> [suffix]
set -u
tag="$1"; dir="$2"; sfx="${3:-final}"
cd "$(dirname "$0")/.."
t=gpurun_out/$tag
cp ${t}_synth4096/pmc_kernels.json profiles/pmc_kernels_synth4096.json
cp ${t}_synth4096/pmc_kernels.json $dir/pmc_kernels_synth4096_$sfx.json
cp ${t}_synth4096/kernel_stats.csv $dir/kernel_stats_synth4096_$sfx.csv
cp ${t}_synth4096/bench.json $dir/bench_synth4096_$sfx.json
cp ${t}_pytest.log $dir/pytest_gpu_$sfx.log
for w in 2 4 8; do cp gpurun_out/rank_synth8192_w$w/pmc_kernels.json profiles/pmc_kernels_synth8192_w$w.json; cp gpurun_out/rank_synth8192_w$w/kernel_stats.csv $dir/kernel_stats_synth8192_rank0_of_$w.csv; done
cp ${t}_emulate_synth8192.jsonl $dir/emulate_synth8192_$sfx.jsonl
cp ${t}_scenes.jsonl $dir/scenes_$sfx.jsonl
cp ${t}_bench_material4096.json $dir/bench_material4096_$sfx.json; cp ${t}_bench_icons4096.json $dir/bench_icons4096_$sfx.json
cp ${t}_rehearse4.json $dir/rehearse_gloo4_one_gpu.json
cp ${t}_timeline_tile.txt $dir/timeline_tile_kernel_$sfx.txt
grep "pb stamp" ${t}_pb_stamp.txt > $dir/pb_stamp_$sfx.txt
cp gpurun_out/contract_counts.jsonl $dir/contract_counts.jsonl 2>/dev/null
python3 - <<P
import json, sys
sys.path.insert(0, ".")
import bench
d = json.loads(open("$dir/bench_synth4096_$sfx.json").read().strip().splitlines()[-1])
r = d["roofline"]
print("bench", d["value"], d["ms_per_step"], "parity bad", d["parity"]["bad"], "tile", r["avg_launch_ms"], "geo", r["geometry_ms"], "frac", r["frac"])
for f in ("synth4096", "synth8192_w2", "synth8192_w4", "synth8192_w8"):
    k = json.load(open("profiles/pmc_kernels_%s.json" % f))
    print(f, "counters match the source:", k.get("source_sha256") == bench.source_sha256())
P
