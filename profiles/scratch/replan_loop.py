#!/usr/bin/env python3
"""N frames with new geometry (set_transforms + svgr_batch_draw) or N cold frames of the bench scene: the thing a kernel trace is taken of.
   python3 profiles/scratch/replan_loop.py replan|cold [n]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

import bench  # noqa: E402
import svgrasterize_amd as S  # noqa: E402
from svgrasterize_amd import _abi  # noqa: E402

mode = sys.argv[1] if len(sys.argv) > 1 else "replan"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 40
sc, _ = bench.load_workload("synth4096")
ctx = S.Context.get(0)
rows, cols = int(sc["viewport"][2]), int(sc["viewport"][3])
out = ctx.alloc(rows * cols * 16)
flags = _abi.RENDER_CLIP01


def new():
    return _abi.Batch(ctx, sc["segs"], sc["seg_kind"], sc["path_seg_off"], sc["path_m6"], sc["path_rule"], sc["path_paint"], viewport=sc["viewport"])


b = new()
b.draw(out, _abi.OUT_CANVAS_F32, flags)
m6 = np.array(sc["path_m6"], dtype=np.float64, copy=True)
ts = []
for i in range(n):
    if mode == "replan":
        m = m6.copy()
        m[:, 2] += 0.125 * (i + 1)
        m[:, 5] += 0.0625 * (i + 1)
        ctx.sync()
        t0 = time.perf_counter()
        b.set_transforms(m)
        t1 = time.perf_counter()
        b.draw(out, _abi.OUT_CANVAS_F32, flags)
        t2 = time.perf_counter()
        ts.append((t1 - t0, t2 - t1))
    else:
        b.destroy()
        ctx.sync()
        t0 = time.perf_counter()
        b = new()
        t1 = time.perf_counter()
        b.draw(out, _abi.OUT_CANVAS_F32, flags)
        t2 = time.perf_counter()
        ts.append((t1 - t0, t2 - t1))
a = np.array(ts[5:]) * 1e3
print(f"{mode}: first call {a[:, 0].mean():.4f} ms (set_transforms / create), draw {a[:, 1].mean():.4f} ms, frame {a.sum(axis=1).mean():.4f} ms (min {a.sum(axis=1).min():.4f})")
