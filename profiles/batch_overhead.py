#!/usr/bin/env python3
"""Host-side cost of one small batch (the per-node route pays it once per Path.mask / Path.fill / solid run):
create, plan, bboxes read-back, render, sync.  python profiles/batch_overhead.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import svgrasterize_amd as S
from svgrasterize_amd import _abi

ctx = S.Context.get(0)
path = S.Path.from_svg("M10,30 C10,5 40,5 40,30 S70,55 40,60 C20,62 10,50 10,30 Z M80,80 h30 v30 h-30 z")
segs, kinds = path.packed()
n = 12
segs = np.tile(segs, (n, 1)); kinds = np.tile(kinds, n)
per = len(kinds) // n
off = np.arange(n + 1) * per
m6 = np.tile(np.array([0.0, 1, 0, 1, 0, 0]), (n, 1)); rule = np.zeros(n, np.uint8); paint = np.tile([0.2, 0.3, 0.1, 0.5], (n, 1))
out = ctx.alloc(256 * 256 * 16)
T = dict(create=0.0, plan=0.0, bboxes=0.0, render=0.0, sync=0.0, destroy=0.0)
R = 300
for it in range(R + 20):
    t = [time.perf_counter()]
    b = _abi.Batch(ctx, segs, kinds, off, m6, rule, paint, viewport=[0, 0, 256, 256]); t.append(time.perf_counter())
    b.plan(); t.append(time.perf_counter())
    b.bboxes(); t.append(time.perf_counter())
    b.render(out, _abi.OUT_CANVAS_F32); t.append(time.perf_counter())
    ctx.sync(); t.append(time.perf_counter())
    b.destroy(); t.append(time.perf_counter())
    if it >= 20:
        for k, (a, c) in zip(T, zip(t, t[1:])):
            T[k] += c - a
print({k: round(v / R * 1e6, 1) for k, v in T.items()}, "us per call; total", round(sum(T.values()) / R * 1e6, 1))
