#!/bin/bash
# Everything the round's record holds, in one call on the GPU box:  profiles/collect_round.sh <tag>
#   GPU test suite, bench line + kernel stats + counters of the N = 1 workload (collect2.sh), counters of rank 0 of the 2 / 4 / 8-way
#   shardings of config 4 (collect_rank.sh), the 1 / 2 / 4 / 8-rank emulation, the real-asset scenes, a 4-rank gloo rehearsal of
#   bench.py --gpus 4 on this one GPU, per-workgroup timelines of the two large kernels.   -> gpurun_out/<tag>_*
set -u
tag="${1:-round}"
cd /tmp; export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
python -m pytest tests -m gpu -x -q > gpurun_out/${tag}_pytest.log 2>&1; tail -2 gpurun_out/${tag}_pytest.log
bash profiles/collect2.sh ${tag}_synth4096 synth4096 > gpurun_out/${tag}_collect2.log 2>&1 || echo "collect2 failed"
for w in 2 4 8; do bash profiles/collect_rank.sh synth8192 $w > gpurun_out/${tag}_rank_w$w.log 2>&1 || echo "collect_rank $w failed"; done
for w in 1 2 4 8; do python profiles/emulate_rank.py --world $w --all --workload synth8192 --steps 40 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(json.dumps(dict(world=d['world'], strip_bands=d['strip_bands'], slowest_ms=d['slowest']['ms_step'], ranks_ms=[r['ms_step'] for r in d['ranks']])))"; done > gpurun_out/${tag}_emulate_synth8192.jsonl
python bench_scenes.py --repeat 5 > gpurun_out/${tag}_scenes.jsonl 2> gpurun_out/${tag}_scenes.err
for wl in material4096 icons4096; do python bench.py --workload $wl 2> /dev/null | tail -1 > gpurun_out/${tag}_bench_$wl.json; done
HSA_ENABLE_IPC_MODE_LEGACY=0 SVGR_BENCH_BACKEND=gloo SVGR_BENCH_DEVICE=0 timeout -k 10 400 python bench.py --gpus 4 --steps 20 --warmup 3 2> gpurun_out/${tag}_rehearse4.err | tail -1 > gpurun_out/${tag}_rehearse4.json   # (bench.py starts its own four ranks)
bash profiles/timeline.sh > gpurun_out/${tag}_timeline.log 2>&1; cp gpurun_out/timeline.txt gpurun_out/${tag}_timeline_tile.txt
bash profiles/pb_stamp.sh > gpurun_out/${tag}_pb_stamp.txt 2>&1
echo "--- bench"; python - <<P
import json
d = json.loads(open("gpurun_out/${tag}_synth4096/bench.json").read().strip().splitlines()[-1])
r = d["roofline"]
print(d["value"], d["ms_per_step"], "parity bad", d["parity"]["bad"], "tile", r["avg_launch_ms"], "geo", r["geometry_ms"], "frac", r["frac"], "stale", r["counters"]["stale"])
P
cat gpurun_out/${tag}_emulate_synth8192.jsonl
cut -c1-160 gpurun_out/${tag}_scenes.jsonl
grep "pb stamp" gpurun_out/${tag}_pb_stamp.txt
