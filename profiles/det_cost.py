#!/usr/bin/env python3
"""What SVGR_RENDER_DETERMINISTIC costs on the bench scene: the step with and without the flag (one wave scatters, in list order;
k_path_build's first wave alone takes the rows).    python profiles/det_cost.py [steps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import svgrasterize_amd as S
from svgrasterize_amd import _abi, synth

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 60
ctx = S.Context.get(0)
sc = synth.make_scene(4096, 4096)
b = _abi.Batch(ctx, sc["segs"], sc["seg_kind"], sc["path_seg_off"], sc["path_m6"], sc["path_rule"], sc["path_paint"], viewport=sc["viewport"])
b.plan()
out = ctx.alloc(4096 * 4096 * 16)
res = {}
for name, flags in (("default", _abi.RENDER_CLIP01), ("deterministic", _abi.RENDER_CLIP01 | _abi.RENDER_DETERMINISTIC)):
    for _ in range(5):
        b.render(out, _abi.OUT_CANVAS_F32, flags)
    ctx.sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        b.render(out, _abi.OUT_CANVAS_F32, flags)
    ctx.sync()
    res[name] = (time.perf_counter() - t0) / steps * 1e3
a = out.download((4096, 4096, 4), np.float32)
b.render(out, _abi.OUT_CANVAS_F32, _abi.RENDER_CLIP01 | _abi.RENDER_DETERMINISTIC)
c = out.download((4096, 4096, 4), np.float32)
print({"ms_per_step": {k: round(v, 4) for k, v in res.items()}, "ratio": round(res["deterministic"] / res["default"], 2),
       "two_deterministic_renders_bit_identical": bool(np.array_equal(a, c))})
