#!/bin/bash
# copy the re-collected counters / traces of a `r06fin` gpurun call into the tracked record:  profiles/publish_r06.sh
set -u
cd "$(dirname "$0")/.."
d=profiles/r06
for wl in synth4096 material4096 icons4096; do
  cp gpurun_out/r06fin_$wl/pmc_kernels.json profiles/pmc_kernels_$wl.json
  cp gpurun_out/r06fin_$wl/pmc_kernels.json $d/pmc_kernels_${wl}_final.json
  cp gpurun_out/r06fin_$wl/kernel_stats.csv $d/kernel_stats_${wl}_final.csv
  cp gpurun_out/r06fin_$wl/bench.json $d/bench_${wl}_final.json
done
for w in 2 4 8; do cp gpurun_out/rank_synth8192_w$w/pmc_kernels.json profiles/pmc_kernels_synth8192_w$w.json; cp gpurun_out/rank_synth8192_w$w/kernel_stats.csv $d/kernel_stats_synth8192_rank0_of_$w.csv; done
cp gpurun_out/replan_trace_replan.txt $d/replan_trace_replan_final.txt; cp gpurun_out/replan_trace_cold.txt $d/replan_trace_cold_final.txt
[ -f gpurun_out/r06fin_pytest.log ] && cp gpurun_out/r06fin_pytest.log $d/pytest_gpu_final.log
python3 - <<P
import json, sys
sys.path.insert(0, ".")
import bench
for f in ("synth4096", "material4096", "icons4096", "synth8192_w2", "synth8192_w4", "synth8192_w8"):
    k = json.load(open("profiles/pmc_kernels_%s.json" % f))
    print(f, "counters match the source:", k.get("source_sha256") == bench.source_sha256())
d = json.loads(open("$d/bench_synth4096_final.json").read().strip().splitlines()[-1])
r = d["roofline"]
print({x: d.get(x) for x in ("value", "ms_per_step", "cold_ms", "cold_fresh_memory_ms", "replan_ms", "value_replan")}, "frac", r["frac"], "tile", r["avg_launch_ms"], "traffic", r["traffic"], "stale", r["counters"]["stale"])
print("parity", d["parity"]["bad"], d["parity_replan"]["bad"], [(c["config"], c.get("ms_per_step"), c.get("warm_ms"), c.get("device_ms"), (c.get("parity") or {}).get("bad")) for c in d["configs"]])
for wl in ("material4096", "icons4096"):
    e = json.loads(open("$d/bench_%s_final.json" % wl).read().strip().splitlines()[-1])
    print(wl, e["ms_per_step"], e["cold_ms"], e["parity"]["bad"], e["roofline"]["counters"]["stale"])
P
