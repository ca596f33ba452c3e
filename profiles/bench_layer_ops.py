#!/usr/bin/env python3
"""Streaming rate of the per-node layer kernels on a 4096 x 4096 RGBA float64 layer (537 MB): wall clock over `reps`
back-to-back launches between two stream syncs, bytes = what the op must read + write.
    python profiles/bench_layer_ops.py [--size 4096] [--reps 20]"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--size", type=int, default=4096)
    ap.add_argument("--reps", type=int, default=20)
    args = ap.parse_args()
    import numpy as np

    import svgrasterize_amd as S
    from svgrasterize_amd import _abi
    from svgrasterize_amd.filters import blur_kernel

    ctx = S.Context.get(0)
    lib = ctx.lib
    n = args.size
    npx = n * n
    bb = (C.c_int64 * 4)(0, 0, n, n)
    rng = np.random.default_rng(1)
    host = rng.random((n, n, 4)) * 0.5
    a, b = ctx.from_host(host), ctx.from_host(host[::-1].copy())
    m1 = ctx.from_host(rng.random((n, n, 1)))
    f32 = ctx.alloc(npx * 16)
    u8 = ctx.alloc(npx * 4)
    tmp = ctx.alloc(npx * 32)
    res = []

    def run(name, nbytes, fn):
        fn()
        ctx.sync()
        t0 = time.perf_counter()
        for _ in range(args.reps):
            fn()
        ctx.sync()
        dt = (time.perf_counter() - t0) / args.reps
        res.append(dict(op=name, ms=round(dt * 1e3, 4), gbs=round(nbytes / dt / 1e9, 1), frac_of_6300=round(nbytes / dt / 6.3e12, 3)))

    px = npx * 32
    run("k_layer_over (4ch src OVER dst)", 3 * px, lambda: _abi._check(lib.svgr_layer_over(ctx.handle, a.handle, bb, b.handle, bb, 4, 0)))
    run("k_layer_over (first: copy)", 2 * px, lambda: _abi._check(lib.svgr_layer_over(ctx.handle, a.handle, bb, b.handle, bb, 4, 1)))
    run("k_layer_in (1ch mask)", 2 * px + npx * 8, lambda: _abi._check(lib.svgr_layer_in(ctx.handle, a.handle, bb, m1.handle, bb, 1)))
    run("k_layer_crop4 (1ch -> 4ch)", px + npx * 8, lambda: _abi._check(lib.svgr_layer_crop4(ctx.handle, tmp.handle, bb, m1.handle, bb, 1)))
    run("k_layer_scale", 2 * px, lambda: _abi._check(lib.svgr_layer_scale(ctx.handle, a.handle, npx * 4, 0.999)))
    run("k_layer_clip01", 2 * px, lambda: _abi._check(lib.svgr_layer_clip01(ctx.handle, a.handle, npx * 4)))
    run("k_layer_convert (pre->straight->sRGB->pre)", 2 * px, lambda: _abi._check(lib.svgr_layer_convert(ctx.handle, a.handle, npx, 1 | 4 | 8)))
    run("k_to_f32", px + npx * 16, lambda: _abi._check(lib.svgr_layer_to_f32(ctx.handle, f32.handle, a.handle, npx * 4, 1)))
    run("k_to_rgba8", px + npx * 4, lambda: _abi._check(lib.svgr_layer_to_rgba8(ctx.handle, u8.handle, a.handle, npx)))
    # gradient fill over the whole layer (linear, 8 stops) times a mask
    stops = [(i / 7.0, np.array([0.1 * i, 0.05 * i, 0.5, 0.9])) for i in range(8)]
    g = S.GradLinear(np.array([0.0, 0.0]), np.array([float(n), float(n)]), stops, None, "pad", False, None)
    gs, keep = g.abi(S.Transform(), False)
    run("k_gradient_fill (linear, 8 stops, x mask)", px + npx * 8,
        lambda: _abi._check(lib.svgr_gradient_fill(ctx.handle, C.byref(gs), m1.handle, bb, tmp.handle)))
    gr = S.GradRadial(np.array([n / 2.0, n / 2.0]), n / 2.0, np.array([n / 2.2, n / 2.1]), 10.0, stops, None, "reflect", False, None)
    gs2, keep2 = gr.abi(S.Transform(), False)
    run("k_gradient_fill (focal radial, reflect, 8 stops)", 2 * (px + npx * 8),   # (+ the det<0 pass reads the grid too)
        lambda: _abi._check(lib.svgr_gradient_fill(ctx.handle, C.byref(gs2), m1.handle, bb, tmp.handle)))
    # separable blur, 25 x 25 taps, on a 2048 x 2048 layer (two passes: read + write each)
    k = blur_kernel(S.Transform().matrix(0, 1, 0, 1, 0, 0).scale(3.0), (1.6, 1.6))
    m = n // 2
    src = ctx.from_host(host[:m, :m].copy())
    out = ctx.alloc((m + k.shape[0] - 1) * (m + k.shape[1] - 1) * 32)
    kk = np.ascontiguousarray(k)
    run(f"svgr_layer_convolve separable {k.shape[0]}x{k.shape[1]} on {m}x{m}", 4 * m * m * 32,
        lambda: _abi._check(lib.svgr_layer_convolve(ctx.handle, out.handle, src.handle, m, m, kk.ctypes.data_as(_abi._P), k.shape[0], k.shape[1])))
    for r in res:
        print(json.dumps(r))


if __name__ == "__main__":
    main()
