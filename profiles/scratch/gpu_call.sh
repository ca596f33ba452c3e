set -u
python -m pytest tests -m gpu -x -q 2>&1 | tail -4
python bench.py --steps 100 2>/dev/null | tail -1 > gpurun_out/bench_now.json
python - <<'P'
import json
d=json.loads(open("gpurun_out/bench_now.json").read())
r=d["roofline"]
print("bench: step", d["ms_per_step"], "tile", r["avg_launch_ms"], "geo", r["geometry_ms"], "frac", r["frac"], "parity bad", d["parity"]["bad"], "cold", d.get("cold_ms"), "replan", d.get("replan_ms"), "plan", d.get("plan_ms"), "x", d.get("replan_over_step"))
P
HSA_ENABLE_IPC_MODE_LEGACY=0 SVGR_BENCH_BACKEND=gloo SVGR_BENCH_DEVICE=0 timeout -k 10 500 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 10 --warmup 2 2> gpurun_out/rehearse2.err | tail -1 > gpurun_out/rehearse2.json
python - <<'P'
import json
d=json.loads(open("gpurun_out/rehearse2.json").read())
print("2-rank gloo rehearsal: step", d["ms_per_step"], "parity", d.get("parity"), "cpu", {k: d.get("cpu_baseline", {}).get(k) for k in ("value","cores","sample")})
P
