#!/usr/bin/env python3
"""What a cold render of the bench scene is made of (host clock, each stage behind its own sync):
   python3 profiles/cold_breakdown.py [synth4096]   -> one JSON line per round"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import svgrasterize_amd as S  # noqa: E402
from svgrasterize_amd import _abi  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "synth4096"
sc, desc = bench.load_workload(name)
ctx = S.Context.get(0)
rows, cols = int(sc["viewport"][2]), int(sc["viewport"][3])
out = ctx.alloc(rows * cols * 16)
for rnd in range(6):
    ctx.sync()
    t0 = time.perf_counter()
    b = _abi.Batch(ctx, sc["segs"], sc["seg_kind"], sc["path_seg_off"], sc["path_m6"], sc["path_rule"], sc["path_paint"], viewport=sc["viewport"])
    t1 = time.perf_counter()
    ctx.sync()
    t2 = time.perf_counter()
    b.plan()
    t3 = time.perf_counter()
    b.render(out, _abi.OUT_CANVAS_F32, _abi.RENDER_CLIP01)
    t4 = time.perf_counter()
    ctx.sync()
    t5 = time.perf_counter()
    b.render(out, _abi.OUT_CANVAS_F32, _abi.RENDER_CLIP01)
    ctx.sync()
    t6 = time.perf_counter()
    b.destroy()
    t7 = time.perf_counter()
    print(json.dumps({"round": rnd, "create_ms": round((t1 - t0) * 1e3, 3), "upload_drain_ms": round((t2 - t1) * 1e3, 3), "plan_ms": round((t3 - t2) * 1e3, 3),
                      "first_render_issue_ms": round((t4 - t3) * 1e3, 3), "first_render_drain_ms": round((t5 - t4) * 1e3, 3),
                      "second_render_ms": round((t6 - t5) * 1e3, 3), "destroy_ms": round((t7 - t6) * 1e3, 3),
                      "cold_total_ms": round((t5 - t0) * 1e3, 3)}))
