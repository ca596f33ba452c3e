#!/usr/bin/env python3
"""cProfile of Scene.render for one real-asset scene dump (host-side hot spots of the per-node route).
    python profiles/profile_scene.py icons 3.677"""
import cProfile, os, pstats, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import svgrasterize_amd as S
from svgrasterize_amd import scenedump

name = sys.argv[1] if len(sys.argv) > 1 else "icons"
scale = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
ctx = S.Context.get(0)
swap = S.Transform().matrix(0, 1, 0, 1, 0, 0)
scene, info, z = scenedump.load_scene(os.path.join(ROOT, "tests", "golden", f"scene_{name}.npz"))
h0, w0 = info["size"]
h, w = int(h0 * scale), int(w0 * scale)
tr = swap.scale(scale) if scale != 1.0 else swap
for _ in range(2):
    layer, _ = scene.render(tr, viewport=[0, 0, h, w], linear_rgb=False); layer._device(); ctx.sync()
pr = cProfile.Profile()
t0 = time.perf_counter()
pr.enable()
layer, _ = scene.render(tr, viewport=[0, 0, h, w], linear_rgb=False); layer._device(); ctx.sync()
pr.disable()
print("wall", time.perf_counter() - t0)
pstats.Stats(pr).sort_stats("cumulative").print_stats(28)
pstats.Stats(pr).sort_stats("tottime").print_stats(35)
