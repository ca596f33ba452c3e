#!/usr/bin/env python3
"""Secondary measurements: the real-asset configurations of BASELINE.json through Scene.render
(wall clock, host + device, canvas resident on the GPU at the end).  bench.py remains the contract
benchmark; this script only feeds the table in DESIGN.md.

    python bench_scenes.py [--repeat 3]
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--repeat", type=int, default=3)
    args = ap.parse_args()
    import numpy as np

    import svgrasterize_amd as S
    from svgrasterize_amd import scenedump

    ctx = S.Context.get(0)
    swap = S.Transform().matrix(0, 1, 0, 1, 0, 0)
    cases = [
        ("tiger", "Ghostscript tiger @2048x2048 (182 solid fills, one batch)", 1.0, None),
        ("material", "material-design @4096x4096 (989 fills, 935 clips; per-node + batched runs)", 1.0, None),
        ("icons", "icons.svg @1114x286 native (431 gradients, 65 clips, 37 blurs)", 1.0, None),
        ("icons4096", "icons.svg @4096x1051 (config 5: the reference's loader at width 4096)", 1.0, None),
    ]
    for name, desc, scale, _ in cases:
        scene, info, z = scenedump.load_scene(os.path.join(ROOT, "tests", "golden", f"scene_{name}.npz"))
        h0, w0 = info["size"]
        h, w = int(h0 * scale), int(w0 * scale)
        tr = swap.scale(scale) if scale != 1.0 else swap
        best = None
        for _ in range(args.repeat):
            ctx.sync()
            t0 = time.perf_counter()
            layer, _hull = scene.render(tr, viewport=[0, 0, h, w], linear_rgb=False)
            _dev = layer._device()  # result resident on the device
            ctx.sync()
            dt = time.perf_counter() - t0
            best = dt if best is None else min(best, dt)
        rec = {"scene": desc, "canvas": [h, w], "render_s": round(best, 4), "canvas_mpix_per_s": round(h * w / best / 1e6, 1),
               "render_cache": "off (the default: every render builds and plans its batches; S.set_render_cache(n) keeps them)"}
        if name in ("tiger", "material"):
            # output stage: straight-alpha sRGB + 8-bit quantisation on the device, bytes to the host,
            # then the PNG container (zlib) on the host
            full = S.Layer.compose([S.Layer(np.zeros((h, w, 4)), (0, 0), True, False), layer], S.COMPOSE_OVER, linear_rgb=False)
            full._device()
            ctx.sync()
            t0 = time.perf_counter()
            u8 = full.to_rgba8()
            rec["to_rgba8_s"] = round(time.perf_counter() - t0, 4)
            for lvl in (9, 1):
                t0 = time.perf_counter()
                n = len(S.canvas_to_png(u8, level=lvl).getvalue())
                rec[f"png_zlib{lvl}_s"] = round(time.perf_counter() - t0, 3)
                rec[f"png_zlib{lvl}_bytes"] = n
            threads = min(16, os.cpu_count() or 1)
            t0 = time.perf_counter()
            n = len(S.canvas_to_png(u8, level=9, threads=threads).getvalue())  # same pixels, pieces deflated side by side
            rec["png_zlib9_parallel_s"], rec["png_zlib9_parallel_bytes"], rec["png_threads"] = round(time.perf_counter() - t0, 3), n, threads
        print(json.dumps(rec))


if __name__ == "__main__":
    main()
