#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ from the upstream reference.

TEST INFRASTRUCTURE ONLY; runs only in the build container (needs /root/reference).
The fixtures are DATA: inputs + the outputs the reference produced for them.
No reference source text is stored.  Re-run with:

    python3 oracle/gen_golden.py [--full]      # --full also renders tiger@2048 / material@4096 (minutes)

Reference entry points exercised (file = /root/reference/svgrasterize.py):
    line_signed_coverage          S:2213-2304
    bezier3_flatness/split/flatten_batch   S:2066-2098
    Transform.__call__            S:531-534
    Path.mask / Path.fill         S:922-1019
    Layer.compose / convert / opacity      S:129-207
    Scene.render                  S:649-752
"""
from __future__ import annotations

import argparse
import hashlib
import json
import os
import sys
import zlib

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_loader  # noqa: E402

GOLD = os.path.join(os.path.dirname(HERE), "tests", "golden")
DEMO = os.path.join(ref_loader.REF_DIR, "demo")


def save(name: str, **arrays) -> None:
    path = os.path.join(GOLD, name)
    np.savez_compressed(path, **arrays)
    print(f"  wrote {name}: {os.path.getsize(path) / 1024:.1f} KiB")


# --------------------------------------------------------------------------------------
# A. line_signed_coverage known-answer tests
# --------------------------------------------------------------------------------------
def gen_coverage(ref) -> None:
    rng = np.random.default_rng(0xC0FE)
    cases = []  # (h, w, line(2,2) in (row, col))
    # hand-written corner cases (rows, cols, p0, p1)
    hand = [
        (4, 6, (0.5, 0.25), (3.5, 5.75)),
        (4, 6, (3.5, 5.75), (0.5, 0.25)),  # reversed: exact negation
        (3, 4, (-1.0, -2.5), (2.0, -0.5)),  # entirely left of the canvas -> folds into col 0
        (3, 4, (-1.0, 6.5), (2.0, 9.5)),  # entirely right -> nothing
        (5, 5, (1.0, 1.0), (1.0, 4.0)),  # horizontal: no coverage
        (5, 5, (0.0, 2.0), (5.0, 2.0)),  # vertical on a pixel border
        (5, 5, (0.25, 2.5), (4.75, 2.5)),  # vertical mid pixel
        (5, 5, (-3.0, 2.5), (9.0, 2.5)),  # vertical crossing top and bottom
        (5, 8, (1.2, -3.0), (1.9, 12.0)),  # shallow, one row, crosses both sides
        (5, 8, (1.9, 12.0), (1.2, -3.0)),
        (6, 8, (0.0, 0.0), (6.0, 8.0)),  # diagonal through pixel corners
        (6, 8, (0.0, 8.0), (6.0, 0.0)),
        (6, 8, (2.0, 3.0), (3.0, 4.0)),  # exactly one pixel diagonal
        (6, 8, (2.0, 3.0), (3.0, 5.0)),  # exactly two pixels
        (6, 8, (2.0, 3.0), (3.0, 6.0)),  # exactly three pixels
        (6, 8, (2.5, 7.5), (3.5, 7.9)),  # touches last column
        (6, 8, (2.5, 7.5), (3.5, 8.4)),  # leaves on the right
        (6, 8, (5.5, 1.0), (9.5, 3.0)),  # leaves at the bottom
        (6, 8, (-2.5, 1.0), (0.5, 3.0)),  # enters from the top
        (1, 1, (0.1, 0.2), (0.9, 0.7)),
        (2, 3, (1e-9, 0.5), (2 - 1e-9, 2.5)),
    ]
    for h, w, p0, p1 in hand:
        cases.append((h, w, np.array([p0, p1], dtype=np.float64)))
    # random lines around canvases of different shapes
    for _ in range(400):
        h = int(rng.integers(1, 24))
        w = int(rng.integers(1, 24))
        lo = np.array([-0.5 * h - 2, -0.5 * w - 2])
        hi = np.array([1.5 * h + 2, 1.5 * w + 2])
        line = rng.uniform(lo, hi, size=(2, 2))
        kind = rng.integers(0, 6)
        if kind == 0:  # snap endpoints to the pixel grid
            line = np.round(line)
        elif kind == 1:  # near-vertical
            line[1, 1] = line[0, 1] + rng.uniform(-1e-3, 1e-3)
        elif kind == 2:  # near-horizontal
            line[1, 0] = line[0, 0] + rng.uniform(-0.3, 0.3)
        elif kind == 3:  # half-pixel grid
            line = np.round(line * 2) / 2
        cases.append((h, w, line.astype(np.float64)))
    hs = np.array([c[0] for c in cases], dtype=np.int32)
    ws = np.array([c[1] for c in cases], dtype=np.int32)
    lines = np.stack([c[2] for c in cases])
    traces = []
    for h, w, line in cases:
        canvas = np.zeros((h, w), dtype=np.float64)
        ref.line_signed_coverage(canvas, line)
        traces.append(canvas.ravel())
    offs = np.cumsum([0] + [t.size for t in traces]).astype(np.int64)
    save("coverage_kat.npz", h=hs, w=ws, lines=lines, trace=np.concatenate(traces), trace_off=offs)


# --------------------------------------------------------------------------------------
# B. flatten known-answer tests
# --------------------------------------------------------------------------------------
def gather_defs(ref, path):
    """lines_defs / cubics_defs exactly as Path.mask collects them (S:930-945)."""
    lines, cubics = [], []
    for sub in path.subpaths:
        for seg in sub:
            if seg[0] in ref.PATH_LINES:
                lines.append(np.asarray(seg[1], dtype=np.float64))
            elif seg[0] == ref.PATH_CUBIC:
                cubics.append(np.asarray(seg[1], dtype=np.float64))
            elif seg[0] == ref.PATH_QUAD:
                cubics.append(np.asarray(ref.bezier2_to_bezier3(seg[1]), dtype=np.float64))
            elif seg[0] == ref.PATH_ARC:
                cubics.extend(np.asarray(ref.arc_to_bezier3(*seg[1]), dtype=np.float64))
            else:
                raise ValueError(seg[0])
    lines = np.array(lines, dtype=np.float64).reshape(-1, 2, 2)
    cubics = np.array(cubics, dtype=np.float64).reshape(-1, 4, 2)
    return lines, cubics


def gen_flatten(ref, tiger_scene) -> None:
    rng = np.random.default_rng(0xF1A7)
    batches = {}
    batches["rand_small"] = rng.uniform(-20, 60, size=(64, 4, 2))
    batches["rand_big"] = rng.uniform(-200, 4200, size=(48, 4, 2))
    wiggle = rng.uniform(0, 512, size=(32, 1, 2)) + rng.normal(0, 3.0, size=(32, 4, 2))
    batches["tiny_curves"] = wiggle
    deg = rng.uniform(0, 300, size=(16, 4, 2))
    deg[:8, 1] = deg[:8, 0]
    deg[:8, 2] = deg[:8, 3]  # straight lines as cubics
    deg[8:12, 3] = deg[8:12, 0]  # closed loop cubic
    deg[12:] = deg[12:, :1]  # all four points equal
    batches["degenerate"] = deg
    # real data: first cubics of the tiger at 512 px
    real = []

    def walk(scene, tr):
        t, a = scene
        if t == ref.RENDER_FILL:
            _l, c = gather_defs(ref, a[0])
            if len(c):
                real.append(tr(c))
        elif t == ref.RENDER_GROUP:
            for ch in a:
                walk(ch, tr)
        elif t == ref.RENDER_TRANSFORM:
            walk(a[0], tr @ a[1])
        elif t == ref.RENDER_STROKE:
            pass
        else:
            walk(a[0], tr)

    walk(tiger_scene, ref.Transform().matrix(0, 1, 0, 1, 0, 0).scale(0.25))
    batches["tiger512"] = np.concatenate(real)[:600]
    out = {}
    for name, batch in batches.items():
        batch = np.ascontiguousarray(batch, dtype=np.float64)
        out[name + "_in"] = batch
        out[name + "_flatness"] = ref.bezier3_flatness_batch(batch)
        out[name + "_split"] = ref.bezier3_split_batch(batch)
        out[name + "_edges"] = ref.bezier3_flatten_batch(batch, 0.1)
    # Transform.__call__ vectors
    ms, pts, res = [], [], []
    for _ in range(32):
        m = np.eye(3)
        m[:2, :] = rng.uniform(-3, 3, size=(2, 3))
        p = rng.uniform(-700, 700, size=(9, 4, 2))
        ms.append(m)
        pts.append(p)
        res.append(ref.Transform(m)(p))
    out["tr_m"] = np.stack(ms)
    out["tr_in"] = np.stack(pts)
    out["tr_out"] = np.stack(res)
    # Transform composition (3x3 matmul chain)
    a = ref.Transform().matrix(0, 1, 0, 1, 0, 0).scale(1.7, 0.6).rotate(0.4).translate(3.5, -2.25).skew(0.1, -0.2)
    out["tr_chain"] = a.m
    out["tr_chain_inv"] = a.invert.m
    save("flatten_kat.npz", **out)


# --------------------------------------------------------------------------------------
# C. Path.mask / Path.fill known-answer tests
# --------------------------------------------------------------------------------------
def gen_mask(ref) -> None:
    rng = np.random.default_rng(0xA5C)
    swap = ref.Transform().matrix(0, 1, 0, 1, 0, 0)
    cases = []

    def add(name, d, tr=swap, rule=None, viewport=None, paint=None, linear_rgb=True):
        cases.append(dict(name=name, d=d, tr=tr, rule=rule, viewport=viewport, paint=paint, linear_rgb=linear_rgb))

    add("tri", "M1,1 L5,1 L3,4 Z")
    add("nested_nonzero", "M1,1 H7 V7 H1 Z M3,3 H5 V5 H3 Z")
    add("nested_evenodd", "M1,1 H7 V7 H1 Z M3,3 H5 V5 H3 Z", rule="evenodd")
    add("nested_cw_ccw", "M1,1 H7 V7 H1 Z M3,3 V5 H5 V3 Z")
    add("clip_topleft", "M-3,-3 H4 V4 H-3 Z", viewport=[0, 0, 8, 8])
    add("clip_all_sides", "M-3,-3 H14 V12 H-3 Z", viewport=[2, 1, 6, 7])
    add("clip_empty", "M1,1 H4 V4 H1 Z", viewport=[10, 10, 5, 5])
    add("unclosed", "M2,2 L9,3 L4,8")
    add("quad", "M2,2 Q12,1 10,10 T3,12 Z", tr=swap.scale(1.5))
    add("cubic_blob", "M10,30 C10,5 40,5 40,30 S70,55 40,60 C20,62 10,50 10,30 Z", tr=swap.scale(0.8).translate(3.3, 1.7))
    add("arc_circle", "M20,5 A15,15 0 1 1 19.99,5 Z", tr=swap)
    add("arc_ellipse_rot", "M10,20 A18,9 30 0 1 40,25 L25,40 Z", tr=swap.scale(1.2))
    add("star_nonzero", "M30,2 L47,56 L2,22 L58,22 L13,56 Z")
    add("star_evenodd", "M30,2 L47,56 L2,22 L58,22 L13,56 Z", rule="evenodd")
    add("rot_rect", "M5,5 H40 V25 H5 Z", tr=swap.rotate(0.4).translate(10, 2))
    add("solid_srgb", "M2,2 C30,-5 40,20 20,30 S-5,20 2,2 Z", paint=np.array([0.2, 0.4, 0.1, 0.5]), linear_rgb=False)
    add("solid_linear", "M2,2 C30,-5 40,20 20,30 S-5,20 2,2 Z", paint=np.array([0.2, 0.4, 0.1, 0.5]), linear_rgb=True)
    add("solid_opaque_srgb", "M1,1 H20 V13 H1 Z", paint=np.array([0.001, 0.5, 1.0, 1.0]), linear_rgb=False)
    add("tiny", "M3.2,3.2 L3.3,3.2 L3.3,3.4 Z")
    add("hline_only", "M1,1 H9")
    add("big_coords_clip", "M-500,-300 C2000,-400 900,1800 -200,900 Z", viewport=[16, 16, 48, 40])
    for i in range(12):
        k = int(rng.integers(3, 7))
        pts = rng.uniform(0, 64, size=(k, 3, 2))
        d = "M{:.3f},{:.3f} ".format(*pts[0, 0])
        for j in range(k):
            c0, c1, p = pts[j]
            d += "C{:.3f},{:.3f} {:.3f},{:.3f} {:.3f},{:.3f} ".format(*c0, *c1, *(pts[(j + 1) % k, 0]))
        d += "Z"
        vp = None
        if i % 3 == 1:
            vp = [int(rng.integers(0, 20)), int(rng.integers(0, 20)), int(rng.integers(8, 40)), int(rng.integers(8, 40))]
        paint = np.concatenate([rng.uniform(0, 1, 3), [1.0]]) * rng.uniform(0.2, 1.0) if i % 2 else None
        add(f"rand{i}", d, rule="evenodd" if i % 4 == 3 else None, viewport=vp, paint=paint, linear_rgb=bool(i % 4 == 1))

    out = {}
    meta = []
    for idx, c in enumerate(cases):
        path = ref.Path.from_svg(c["d"])
        lines, cubics = gather_defs(ref, path)
        if c["paint"] is None:
            res = path.mask(c["tr"], fill_rule=c["rule"], viewport=c["viewport"])
        else:
            res = path.fill(c["tr"], c["paint"], fill_rule=c["rule"], viewport=c["viewport"], linear_rgb=c["linear_rgb"])
        m = dict(
            name=c["name"], d=c["d"], rule=c["rule"], viewport=c["viewport"], linear_rgb=c["linear_rgb"],
            has_paint=c["paint"] is not None, none=res is None,
        )
        out[f"{idx}_tr"] = c["tr"].m
        out[f"{idx}_lines"] = lines
        out[f"{idx}_cubics"] = cubics
        # segment list (type, points) so the host Path can be built without an SVG parser
        segt, segp, subs = [], [], []
        for sub in path.subpaths:
            subs.append(len(sub))
            for seg in sub:
                segt.append(seg[0])
                if seg[0] == ref.PATH_ARC:
                    center, rx, ry, phi, eta, eta_delta = seg[1]
                    segp.append(np.array([center[0], center[1], rx, ry, phi, eta, eta_delta, 0.0]))
                else:
                    p = np.asarray(seg[1], dtype=np.float64).ravel()
                    segp.append(np.concatenate([p, np.zeros(8 - p.size)]))
        out[f"{idx}_segt"] = np.array(segt, dtype=np.int32)
        out[f"{idx}_segp"] = np.array(segp, dtype=np.float64).reshape(-1, 8)
        out[f"{idx}_subs"] = np.array(subs, dtype=np.int32)
        if c["paint"] is not None:
            out[f"{idx}_paint"] = c["paint"]
        if res is not None:
            layer, hull = res
            m["offset"] = [int(layer.offset[0]), int(layer.offset[1])]
            m["pre_alpha"] = bool(layer.pre_alpha)
            m["layer_linear_rgb"] = bool(layer.linear_rgb)
            out[f"{idx}_image"] = layer.image
            out[f"{idx}_hull"] = np.array(hull.points, dtype=np.float64)
        meta.append(m)
    out["meta"] = np.array(json.dumps(meta))
    save("mask_kat.npz", **out)


# --------------------------------------------------------------------------------------
# D. Layer.compose / convert / opacity
# --------------------------------------------------------------------------------------
def gen_compose(ref) -> None:
    rng = np.random.default_rng(0xC0A1)
    out = {}
    meta = []

    def rnd_layer(ch, pre_alpha=True, linear_rgb=False):
        r, c = int(rng.integers(3, 14)), int(rng.integers(3, 14))
        if ch == 1:
            img = rng.uniform(0, 1, size=(r, c, 1))
            img[rng.uniform(size=img.shape) < 0.3] = 0.0
            img[rng.uniform(size=img.shape) < 0.2] = 1.0
        else:
            a = rng.uniform(0, 1, size=(r, c, 1))
            a[rng.uniform(size=a.shape) < 0.25] = 0.0
            a[rng.uniform(size=a.shape) < 0.25] = 1.0
            rgb = rng.uniform(0, 1, size=(r, c, 3))
            img = np.concatenate([rgb * a if pre_alpha else rgb, a], axis=-1)
        off = (int(rng.integers(-6, 10)), int(rng.integers(-6, 10)))
        return ref.Layer(img, off, pre_alpha, linear_rgb)

    def record(tag, layers, result, **kw):
        idx = len(meta)
        m = dict(tag=tag, n=len(layers), none=result is None, **kw)
        m["in"] = [dict(offset=list(map(int, l.offset)), pre_alpha=bool(l.pre_alpha), linear_rgb=bool(l.linear_rgb)) for l in layers]
        for j, l in enumerate(layers):
            out[f"{idx}_in{j}"] = l.image
        if result is not None:
            m["offset"] = list(map(int, result.offset))
            m["pre_alpha"] = bool(result.pre_alpha)
            m["linear_rgb"] = bool(result.linear_rgb)
            out[f"{idx}_out"] = result.image
        meta.append(m)

    for trial in range(10):
        n = int(rng.integers(2, 6))
        layers = [rnd_layer(4 if rng.uniform() < 0.8 else 1) for _ in range(n)]
        record("over", layers, ref.Layer.compose(layers, ref.COMPOSE_OVER, False), method=0, linear_rgb=False)
    for trial in range(6):  # mixed colour spaces: forces convert()
        layers = [rnd_layer(4, pre_alpha=bool(rng.integers(0, 2)), linear_rgb=bool(rng.integers(0, 2))) for _ in range(3)]
        lin = bool(trial % 2)
        record("over_convert", layers, ref.Layer.compose(layers, ref.COMPOSE_OVER, lin), method=0, linear_rgb=lin)
    for trial in range(10):
        mask = rnd_layer(1, True, True)
        img = rnd_layer(4)
        layers = [mask, img]
        record("in", layers, ref.Layer.compose(layers, ref.COMPOSE_IN, False), method=2, linear_rgb=False)
    # disjoint IN -> None
    a = ref.Layer(np.ones((3, 3, 1)), (0, 0), True, True)
    b = ref.Layer(np.ones((3, 3, 4)), (10, 10), True, False)
    record("in_empty", [a, b], ref.Layer.compose([a, b], ref.COMPOSE_IN, False), method=2, linear_rgb=False)
    for method in (ref.COMPOSE_OUT, ref.COMPOSE_ATOP, ref.COMPOSE_XOR):
        layers = [rnd_layer(4), rnd_layer(4)]
        record("full", layers, ref.Layer.compose(layers, method, False), method=int(method), linear_rgb=False)
    for trial in range(6):
        l = rnd_layer(4, pre_alpha=bool(trial % 2), linear_rgb=bool((trial // 2) % 2))
        record("convert", [l], l.convert(pre_alpha=bool((trial + 1) % 2), linear_rgb=bool(trial % 3 == 0)),
               to_pre_alpha=bool((trial + 1) % 2), to_linear_rgb=bool(trial % 3 == 0))
    for trial in range(4):
        l = rnd_layer(4, pre_alpha=bool(trial % 2), linear_rgb=False)
        record("opacity", [l], l.opacity(0.37 + 0.1 * trial, linear_rgb=bool(trial // 2)), opacity=0.37 + 0.1 * trial,
               linear_rgb=bool(trial // 2))
    out["meta"] = np.array(json.dumps(meta))
    save("compose_kat.npz", **out)


# --------------------------------------------------------------------------------------
# D2. gradient fills (S:1021-1047, 1544-1695) and Gaussian blur (S:106-118, 1890-1944)
# --------------------------------------------------------------------------------------
def gen_gradient(ref) -> None:
    rng = np.random.default_rng(0x96AD)
    swap = ref.Transform().matrix(0, 1, 0, 1, 0, 0)
    out, meta = {}, []

    def stops(n, alpha=True):
        offs = np.sort(rng.uniform(0, 1, n))
        offs[0] = 0.0 if rng.uniform() < 0.5 else offs[0]
        offs[-1] = 1.0 if rng.uniform() < 0.5 else offs[-1]
        res = []
        for o in offs:
            a = rng.uniform(0.2, 1.0) if alpha else 1.0
            res.append((float(o), np.concatenate([rng.uniform(0, 1, 3) * a, [a]])))
        return res

    shapes = ["M4,4 H60 V44 H4 Z", "M32,3 C70,5 62,50 30,46 S-6,30 32,3 Z", "M8,8 L56,12 L40,44 L10,36 Z"]
    cases = []
    for i in range(18):
        d = shapes[i % 3]
        tr = swap.scale(rng.uniform(0.6, 1.6)).translate(rng.uniform(-3, 3), rng.uniform(-3, 3))
        if i % 5 == 4:
            tr = tr.rotate(0.3)
        gt = None if i % 3 == 0 else ref.Transform().translate(rng.uniform(-5, 5), rng.uniform(-5, 5)).scale(
            rng.uniform(0.7, 1.4), rng.uniform(0.7, 1.4)).rotate(rng.uniform(-0.5, 0.5))
        spread = ["pad", "repeat", "reflect"][i % 3]
        bbox_units = i % 4 == 3
        lin = [None, True, False][i % 3]
        if i % 2 == 0:
            if bbox_units:
                p0, p1 = np.array([0.1, 0.2]), np.array([0.7, 0.9])
            else:
                p0, p1 = rng.uniform(5, 30, 2), rng.uniform(30, 60, 2)
            paint = ref.GradLinear(p0, p1, stops(int(rng.integers(2, 6))), gt, spread, bbox_units, lin)
            kind = "linear"
        else:
            if bbox_units:
                c, r = np.array([0.5, 0.45]), 0.4
                fc = None if i % 3 == 0 else np.array([0.6, 0.5])
            else:
                c, r = rng.uniform(20, 40, 2), float(rng.uniform(10, 30))
                fc = None if i % 3 == 0 else c + rng.uniform(-0.5, 0.5, 2) * r
            fr = None if fc is None else (0.0 if i % 4 else 0.2 * r)
            if i == 13:  # focal point outside the circle: det < 0 somewhere
                fc = c + np.array([1.3 * r, 0.0])
            paint = ref.GradRadial(c, r, fc, fr, stops(int(rng.integers(2, 6))), gt, spread, bbox_units, lin)
            kind = "radial"
        vp = [0, 0, 70, 90] if i % 6 == 5 else None
        cases.append((d, tr, paint, kind, vp, bool(i % 2)))

    for idx, (d, tr, paint, kind, vp, linear_rgb) in enumerate(cases):
        path = ref.Path.from_svg(d)
        lines, cubics = gather_defs(ref, path)
        res = path.fill(tr, paint, viewport=vp, linear_rgb=linear_rgb)
        layer, _hull = res
        m = dict(kind=kind, viewport=vp, linear_rgb=linear_rgb, spread=paint.spread, bbox_units=bool(paint.bbox_units),
                 paint_linear_rgb=paint.linear_rgb, offset=[int(layer.offset[0]), int(layer.offset[1])],
                 layer_linear_rgb=bool(layer.linear_rgb), has_gt=paint.transform is not None)
        out[f"{idx}_tr"] = tr.m
        out[f"{idx}_lines"] = lines
        out[f"{idx}_cubics"] = cubics
        out[f"{idx}_stop_off"] = np.array([o for o, _ in paint.stops])
        out[f"{idx}_stop_col"] = np.array([c for _, c in paint.stops])
        if paint.transform is not None:
            out[f"{idx}_gt"] = paint.transform.m
        if kind == "linear":
            out[f"{idx}_p0"], out[f"{idx}_p1"] = np.asarray(paint.p0, float), np.asarray(paint.p1, float)
        else:
            out[f"{idx}_center"] = np.asarray(paint.center, float)
            out[f"{idx}_radius"] = np.array(float(paint.radius))
            m["has_focal"] = paint.fcenter is not None
            if paint.fcenter is not None:
                out[f"{idx}_fcenter"] = np.asarray(paint.fcenter, float)
                out[f"{idx}_fradius"] = np.array(float(paint.fradius))
        out[f"{idx}_image"] = layer.image
        # Grad*.fill on explicit coordinates (S:1553, S:1577): user-space points around the shape, both colour spaces
        pts = np.random.default_rng(1000 + idx).uniform(-0.2, 1.2, (9, 7, 2)) * ([1.0, 1.0] if paint.bbox_units else [70.0, 50.0])
        out[f"{idx}_eval_pts"] = pts
        out[f"{idx}_eval_lin"] = paint.fill(pts, linear_rgb=True)
        out[f"{idx}_eval_srgb"] = paint.fill(pts, linear_rgb=False)
        meta.append(m)

    # blur kernels + convolutions
    bl = []
    for j in range(8):
        tr = swap.scale(rng.uniform(0.8, 3.0))
        if j % 4 == 3:
            tr = tr.rotate(0.4)
        if j == 5:
            tr = swap.scale(2.0, 0.7)
        sig = (float(rng.uniform(0.3, 2.5)), None if j % 2 else float(rng.uniform(0.3, 2.5)))
        kernel = ref.blur_kernel(tr, (sig[0], sig[0] if sig[1] is None else sig[1]))
        a = rng.uniform(0, 1, (int(rng.integers(5, 20)), int(rng.integers(5, 20)), 1))
        img = np.concatenate([rng.uniform(0, 1, a.shape[:2] + (3,)) * a, a], axis=-1)
        layer = ref.Layer(img, (int(rng.integers(-5, 9)), int(rng.integers(-5, 9))), True, bool(j % 2))
        res = ref.filter_blur(tr, *sig)(layer)
        out[f"b{j}_tr"] = tr.m
        out[f"b{j}_in"] = img
        out[f"b{j}_out"] = res.image
        if kernel is not None:
            out[f"b{j}_kernel"] = kernel
        bl.append(dict(sigma=[sig[0], sig[1]], in_offset=[int(v) for v in layer.offset], in_linear_rgb=bool(layer.linear_rgb),
                       noop=kernel is None, out_offset=[int(v) for v in res.offset], out_pre_alpha=bool(res.pre_alpha),
                       out_linear_rgb=bool(res.linear_rgb)))
    # a degenerate blur (both sigmas below half a pixel): no-op
    out["meta"] = np.array(json.dumps(dict(grad=meta, blur=bl)))
    save("gradient_blur_kat.npz", **out)


# --------------------------------------------------------------------------------------
# E. scene dumps + golden renders
# --------------------------------------------------------------------------------------
class Dumper:
    """Serialise a reference Scene into plain data (tree JSON + geometry arrays).

    STROKE nodes are stored as FILL nodes of the reference-stroked path (the stroker is
    out of scope; ``path.stroke`` runs before the transform, S:668, so it is
    resolution independent)."""

    def __init__(self, ref):
        self.ref = ref
        self.lines, self.cubics = [], []
        self.loff, self.coff = [0], [0]
        self.unsupported = set()

    def add_path(self, path) -> int:
        l, c = gather_defs(self.ref, path)
        self.lines.append(l)
        self.cubics.append(c)
        self.loff.append(self.loff[-1] + len(l))
        self.coff.append(self.coff[-1] + len(c))
        return len(self.loff) - 2

    def paint(self, p):
        ref = self.ref
        if p is None:
            return None
        if isinstance(p, np.ndarray):
            return dict(k="rgba", v=[float(x) for x in p])
        if isinstance(p, ref.GradLinear):
            return dict(k="linear", p0=[float(x) for x in p.p0], p1=[float(x) for x in p.p1],
                        stops=[[float(o), [float(x) for x in c]] for o, c in p.stops],
                        tr=None if p.transform is None else [float(x) for x in p.transform.m[:2].ravel()],
                        spread=p.spread, bbox_units=bool(p.bbox_units), linear_rgb=p.linear_rgb)
        if isinstance(p, ref.GradRadial):
            return dict(k="radial", center=None if p.center is None else [float(x) for x in p.center],
                        radius=None if p.radius is None else float(p.radius),
                        fcenter=None if p.fcenter is None else [float(x) for x in p.fcenter],
                        fradius=None if p.fradius is None else float(p.fradius),
                        stops=[[float(o), [float(x) for x in c]] for o, c in p.stops],
                        tr=None if p.transform is None else [float(x) for x in p.transform.m[:2].ravel()],
                        spread=p.spread, bbox_units=bool(p.bbox_units), linear_rgb=p.linear_rgb)
        if isinstance(p, ref.Pattern):
            return dict(k="pattern", scene=self.node(p.scene), scene_bbox_units=bool(p.scene_bbox_units),
                        scene_view_box=None if p.scene_view_box is None else [float(x) for x in p.scene_view_box],
                        cell=[float(p.x), float(p.y), float(p.width), float(p.height)],
                        tr=[float(x) for x in p.transform.m[:2].ravel()], bbox_units=bool(p.bbox_units))
        self.unsupported.add(type(p).__name__)
        return dict(k="unsupported", name=type(p).__name__)

    def node(self, scene):
        ref = self.ref
        t, a = scene
        if t == ref.RENDER_FILL:
            path, paint, rule = a
            return dict(t="fill", path=self.add_path(path), paint=self.paint(paint), rule=rule)
        if t == ref.RENDER_STROKE:
            path, paint, width, cap, join = a
            return dict(t="fill", path=self.add_path(path.stroke(width, cap, join)), paint=self.paint(paint),
                        rule=None, from_stroke=True)
        if t == ref.RENDER_GROUP:
            return dict(t="group", c=[self.node(ch) for ch in a])
        if t == ref.RENDER_OPACITY:
            return dict(t="opacity", c=self.node(a[0]), o=float(a[1]))
        if t == ref.RENDER_CLIP:
            return dict(t="clip", c=self.node(a[0]), clip=self.node(a[1]), bbox_units=bool(a[2]))
        if t == ref.RENDER_MASK:
            return dict(t="mask", c=self.node(a[0]), mask=self.node(a[1]), bbox_units=bool(a[2]))
        if t == ref.RENDER_TRANSFORM:
            return dict(t="transform", c=self.node(a[0]), m=[float(x) for x in a[1].m[:2].ravel()])
        if t == ref.RENDER_FILTER:
            flt = a[1]
            fl = []
            for ftype, attrs, inputs in flt.filters:
                if ftype != ref.FE_GAUSSIAN_BLUR:
                    self.unsupported.add(f"filter:{ftype}")
                fl.append(dict(type=int(ftype), attrs=[None if v is None else float(v) for v in attrs]
                               if ftype == ref.FE_GAUSSIAN_BLUR else repr(attrs), inputs=[int(i) for i in inputs]))
            return dict(t="filter", c=self.node(a[0]), filters=fl)
        raise ValueError(t)

    def arrays(self):
        cat = lambda xs, shape: np.concatenate(xs) if xs else np.zeros(shape)
        return dict(
            lines=cat(self.lines, (0, 2, 2)), cubics=cat(self.cubics, (0, 4, 2)),
            line_off=np.array(self.loff, dtype=np.int64), cubic_off=np.array(self.coff, dtype=np.int64),
        )


def leaves_with_transform(ref, scene, tr):
    """Yield (path-or-stroked-path, paint, rule, accumulated transform) in paint order."""
    t, a = scene
    if t == ref.RENDER_FILL:
        yield a[0], a[1], a[2], tr
    elif t == ref.RENDER_STROKE:
        yield a[0].stroke(a[2], a[3], a[4]), a[1], None, tr
    elif t == ref.RENDER_GROUP:
        for ch in a:
            yield from leaves_with_transform(ref, ch, tr)
    elif t == ref.RENDER_TRANSFORM:
        yield from leaves_with_transform(ref, a[0], tr @ a[1])
    else:
        yield from leaves_with_transform(ref, a[0], tr)


def sample_pixels(canvas: np.ndarray, rng, n_edge=12000, n_rand=4000):
    """Sparse pin of a big render: flat indices of 'interesting' + random pixels and values."""
    a = canvas[..., 3]
    frac = (a > 1e-6) & (a < 1 - 1e-6)
    idx_edge = np.flatnonzero(frac.ravel())
    if idx_edge.size > n_edge:
        idx_edge = rng.choice(idx_edge, n_edge, replace=False)
    idx_rand = rng.integers(0, a.size, n_rand)
    idx = np.unique(np.concatenate([idx_edge, idx_rand])).astype(np.int64)
    return idx, canvas.reshape(-1, 4)[idx]


def f32_hash(canvas) -> str:
    return hashlib.sha256(np.ascontiguousarray(canvas, dtype=np.float32).tobytes()).hexdigest()


def gen_scene(ref, fonts, name, svg, width, small_scales, full, crop=None, store_layer=True) -> None:
    print(f"scene {name} ({svg} @ {width})")
    rng = np.random.default_rng(zlib.crc32(name.encode()))
    scene, _ids, size = ref.svg_scene_from_filepath(os.path.join(DEMO, svg), width=width, fonts=fonts)
    w, h = size
    d = Dumper(ref)
    tree = d.node(scene)
    out = d.arrays()
    out["tree"] = np.array(json.dumps(tree))
    info = dict(name=name, svg=svg, width=width, size=[int(h), int(w)], unsupported=sorted(d.unsupported), renders=[])
    swap = ref.Transform().matrix(0, 1, 0, 1, 0, 0)
    for scale in small_scales:
        tr = swap.scale(scale)
        hh, ww = int(h * scale), int(w * scale)
        res = scene.render(tr, viewport=[0, 0, hh, ww], linear_rgb=False)
        layer, hull = res
        canvas = np.zeros((hh, ww, 4))
        cl = layer.convert(pre_alpha=True, linear_rgb=False)
        ref.canvas_merge_at(canvas, cl.image, cl.offset)
        tag = f"s{hh}"
        out[f"{tag}_canvas"] = canvas
        if store_layer:
            out[f"{tag}_layer"] = layer.image
        info["renders"].append(dict(tag=tag, scale=scale, size=[hh, ww], layer_offset=list(map(int, layer.offset)),
                                    layer_pre_alpha=bool(layer.pre_alpha), layer_linear_rgb=bool(layer.linear_rgb),
                                    sha256_f32=f32_hash(canvas)))
        # per-leaf integer bboxes (solid scenes only: plain group/transform nesting)
        if not d.unsupported and name == "tiger":
            boxes = []
            for path, paint, rule, ltr in leaves_with_transform(ref, scene, tr):
                r = path.fill(ltr, paint, fill_rule=rule, viewport=[0, 0, hh, ww], linear_rgb=False)
                boxes.append([-1, -1, -1, -1] if r is None else [int(v) for v in r[0].bbox])
            out[f"{tag}_leaf_bbox"] = np.array(boxes, dtype=np.int64)
    if crop is not None:
        # full-resolution crop through the reference's own viewport mechanism (S:968-971)
        res = scene.render(swap, viewport=list(crop), linear_rgb=False)
        if res is not None:
            layer, _ = res
            out["crop_layer"] = layer.image
            info["crop"] = dict(viewport=list(crop), offset=list(map(int, layer.offset)))
    if full:
        res = scene.render(swap, viewport=[0, 0, int(h), int(w)], linear_rgb=False)
        layer, _ = res
        canvas = np.zeros((int(h), int(w), 4))
        cl = layer.convert(pre_alpha=True, linear_rgb=False)
        ref.canvas_merge_at(canvas, cl.image, cl.offset)
        idx, vals = sample_pixels(canvas, rng)
        out["full_idx"] = idx
        out["full_val"] = vals
        info["full"] = dict(size=[int(h), int(w)], sha256_f32=f32_hash(canvas), layer_offset=list(map(int, layer.offset)),
                            layer_shape=list(layer.image.shape))
        print("   full sha256[:16] =", info["full"]["sha256_f32"][:16])
    out["info"] = np.array(json.dumps(info))
    save(f"scene_{name}.npz", **out)



def pack_segments(ref, path):
    """(types, params (n, 8), subpath sizes) of a reference Path: the format Path.from_segments reads."""
    segt, segp, subs = [], [], []
    for sub in path.subpaths:
        subs.append(len(sub))
        for seg in sub:
            segt.append(seg[0])
            if seg[0] == ref.PATH_ARC:
                center, rx, ry, phi, eta, eta_delta = seg[1]
                segp.append(np.array([center[0], center[1], rx, ry, phi, eta, eta_delta, 0.0]))
            else:
                p = np.asarray(seg[1], dtype=np.float64).ravel()
                segp.append(np.concatenate([p, np.zeros(8 - p.size)]))
    return (np.array(segt, dtype=np.int32), np.array(segp, dtype=np.float64).reshape(-1, 8),
            np.array(subs, dtype=np.int32))


def gen_stroke(ref, tiger_scene) -> None:
    """Path.stroke known answers (S:1105-1180): input segments, width, cap, join -> stroked segments.
    Hand-written shapes for every cap x join, degenerate pieces, quads, arcs, the cusp the reference special-cases
    (S:2171), and every STROKE node of the tiger with its own width / cap / join."""
    caps = [None, ref.STROKE_CAP_BUTT, ref.STROKE_CAP_ROUND, ref.STROKE_CAP_SQUARE]
    joins = [None, ref.STROKE_JOIN_MITER, ref.STROKE_JOIN_ROUND, ref.STROKE_JOIN_BEVEL]
    shapes = {
        "polyline": "M10,10 L50,12 L60,40 L20,55",
        "polygon": "M10,10 L50,12 L60,40 L20,55 Z",
        "sharp": "M5,5 L60,8 L8,12",                      # beyond the miter limit
        "collinear": "M0,0 L10,0 L20,0 L20,15",
        "zero_len": "M3,3 L3,3 L9,7 L9,7 L14,2",
        "cubic_s": "M10,80 C40,10 65,10 95,80 S150,150 180,80",
        "cusp": "M0,0 C100,50 0,50 100,0",
        "loop": "M10,10 C90,90 90,10 10,90",
        "quad": "M10,80 Q52.5,10 95,80 T180,80",
        "circle_arcs": "M50,10 A40,40 0 1 1 49.9,10 Z",
        "ellipse_arc": "M10,50 A30,15 30 0 1 70,40",
        "closed_cubic": "M20,20 C60,0 100,40 80,80 C60,120 0,100 20,20 Z",
        "two_subpaths": "M0,0 L30,0 L30,30 Z M50,5 C70,5 70,35 50,35",
        "tiny": "M0,0 L1e-9,0 L5,5",
    }
    cases = []
    for name, d in shapes.items():
        path = ref.Path.from_svg(d)
        for ci, cap in enumerate(caps):
            for ji, join in enumerate(joins):
                if name not in ("polyline", "cubic_s", "two_subpaths") and (ci + ji) % 3 != 0:
                    continue  # the full cap x join grid on three shapes, a diagonal on the rest
                for width in (1.0, 7.5):
                    cases.append((f"{name}/{cap}/{join}/{width}", path, width, cap, join))

    def walk(scene):
        t, a = scene
        if t == ref.RENDER_STROKE:
            yield a
        elif t == ref.RENDER_GROUP:
            for ch in a:
                yield from walk(ch)
        elif t in (ref.RENDER_OPACITY, ref.RENDER_CLIP, ref.RENDER_MASK, ref.RENDER_TRANSFORM, ref.RENDER_FILTER):
            yield from walk(a[0])

    for k, (path, _paint, width, cap, join) in enumerate(walk(tiger_scene.scene if hasattr(tiger_scene, "scene") else tiger_scene)):
        cases.append((f"tiger{k}", path, float(width), cap, join))

    out, meta = {}, []
    for idx, (name, path, width, cap, join) in enumerate(cases):
        it, ip, isub = pack_segments(ref, path)
        res = path.stroke(width, cap, join)
        ot, op, osub = pack_segments(ref, res)
        out[f"{idx}_it"], out[f"{idx}_ip"], out[f"{idx}_is"] = it, ip, isub
        out[f"{idx}_ot"], out[f"{idx}_op"], out[f"{idx}_os"] = ot, op, osub
        meta.append(dict(name=name, width=width, cap=cap, join=join))
    out["meta"] = np.array(json.dumps(meta))
    save("stroke_kat.npz", **out)
    print(f"  stroke: {len(cases)} cases, {sum(len(out[f'{i}_ot']) for i in range(len(cases)))} output segments")



def gen_png(ref) -> None:
    """Output stage known answers (Layer.write_png S:209-213, canvas_to_png S:249-274): layer (image, flags) ->
    the uint8 array the reference hands to zlib and the PNG bytes it writes."""
    rng = np.random.default_rng(20240917)
    out, meta = {}, []
    cases = []
    for k, (shape, pre, lin) in enumerate([((5, 7), True, False), ((16, 9), True, True), ((12, 12), False, False),
                                            ((9, 20), False, True), ((64, 48), True, False)]):
        a = rng.uniform(0, 1, shape + (1,))
        a[rng.uniform(0, 1, shape + (1,)) < 0.2] = 0.0
        rgb = rng.uniform(0, 1, shape + (3,))
        img = np.concatenate([rgb * a if pre else rgb, a], axis=-1)
        if k == 0:
            img[0, 0] = [0.5 / 255 * 0.4, 1.5 / 255 * 0.4, 2.5 / 255 * 0.4, 0.4]  # ties of the rounding after un-premultiply
        cases.append((img, pre, lin))
    for idx, (img, pre, lin) in enumerate(cases):
        layer = ref.Layer(img, (3, 4), pre_alpha=pre, linear_rgb=lin)
        conv = layer.convert(pre_alpha=False, linear_rgb=False)
        u8 = np.round(conv.image * 255.0).astype(np.uint8)
        png = layer.write_png().getvalue()
        out[f"{idx}_image"] = img
        out[f"{idx}_u8"] = u8
        out[f"{idx}_png"] = np.frombuffer(png, dtype=np.uint8)
        meta.append(dict(pre_alpha=pre, linear_rgb=lin))
    out["meta"] = np.array(json.dumps(meta))
    save("png_kat.npz", **out)



def gen_filters(ref) -> None:
    """Remaining compose modes / filter primitives / luminance mask: Layer.compose OUT / ATOP / XOR /
    arithmetic (S:285-297, full-canvas union S:348-365), Layer.color_matrix (S:95-104), Layer.morphology (S:120-127,
    pooling S:419-468), Filter chains with feOffset / feMerge / feBlend / feComposite / feColorMatrix / feMorphology
    (S:1749-1887) and Scene MASK (S:721-741)."""
    rng = np.random.default_rng(77)
    out, meta = {}, dict(compose=[], cmatrix=[], morph=[], chain=[], mask=[])

    def rand_layer(shape, ch, off, pre, lin):
        a = rng.uniform(0, 1, shape + (1,))
        a[rng.uniform(0, 1, shape + (1,)) < 0.25] = 0.0
        if ch == 1:
            img = a
        else:
            rgb = rng.uniform(0, 1, shape + (3,))
            img = np.concatenate([rgb * a if pre else rgb, a], axis=-1)
        return ref.Layer(img, off, pre_alpha=pre, linear_rgb=lin)

    # -- compose modes ------------------------------------------------------------------------------------------
    modes = [ref.COMPOSE_OUT, ref.COMPOSE_ATOP, ref.COMPOSE_XOR, (0.3, 0.5, 0.7, 0.1), (1.0, 0.0, 0.0, 0.0), (0.0, 1.0, -1.0, 0.2)]
    k = 0
    for mode in modes:
        for variant in range(2):
            layers = [rand_layer((9, 7), 4, (2, 3), True, False), rand_layer((6, 11), 4, (5, 0), variant == 0, True)]
            if variant:
                layers.append(rand_layer((8, 8), 1 if not isinstance(mode, tuple) else 4, (0, 6), True, True))
            lin = bool(variant)
            res = ref.Layer.compose(layers, mode, linear_rgb=lin)
            for j, l in enumerate(layers):
                out[f"c{k}_in{j}"] = l.image
            out[f"c{k}_out"] = res.image
            meta["compose"].append(dict(mode=list(mode) if isinstance(mode, tuple) else int(mode), linear_rgb=lin,
                                        layers=[dict(offset=[int(v) for v in l.offset], pre_alpha=l.pre_alpha, linear_rgb=l.linear_rgb)
                                                for l in layers],
                                        out_offset=[int(v) for v in res.offset], out_pre_alpha=res.pre_alpha, out_linear_rgb=res.linear_rgb))
            k += 1

    # -- color matrix --------------------------------------------------------------------------------------------------
    mats = [rng.uniform(-0.5, 1.0, (4, 5)), ref.COLOR_MATRIX_LUM.copy(), np.hstack([np.eye(4), np.zeros((4, 1))])]
    for j, m in enumerate(mats):
        l = rand_layer((10, 13), 4, (4, 1), bool(j % 2), bool(j // 2))
        res = l.color_matrix(m)
        out[f"m{j}_in"], out[f"m{j}_matrix"], out[f"m{j}_out"] = l.image, m, res.image
        meta["cmatrix"].append(dict(offset=[int(v) for v in l.offset], pre_alpha=l.pre_alpha, linear_rgb=l.linear_rgb,
                                    out_pre_alpha=res.pre_alpha, out_linear_rgb=res.linear_rgb))

    # -- morphology --------------------------------------------------------------------------------------------------
    for j, (kx, ky, method) in enumerate([(1, 1, "max"), (3, 2, "min"), (5, 7, "max"), (2, 6, "min")]):
        l = rand_layer((14, 17), 4, (3, 3), bool(j % 2), False)
        res = l.morphology(kx, ky, method)
        out[f"p{j}_in"], out[f"p{j}_out"] = l.image, res.image
        meta["morph"].append(dict(x=kx, y=ky, method=method, offset=[int(v) for v in l.offset], pre_alpha=l.pre_alpha,
                                  linear_rgb=l.linear_rgb, out_offset=[int(v) for v in res.offset],
                                  out_pre_alpha=res.pre_alpha, out_linear_rgb=res.linear_rgb))

    # -- filter chains -----------------------------------------------------------------------------------------------
    swap = ref.Transform().matrix(0, 1, 0, 1, 0, 0)
    chains = [
        ("drop_shadow", lambda f: f.blur(1.2, input=ref.FE_SOURCE_ALPHA, result="b").offset(2.5, 1.5, input="b", result="o")
                                   .merge(["o", ref.FE_SOURCE_GRAPHIC]), swap.scale(1.5)),
        ("composite_ops", lambda f: f.offset(3, -2, result="o").composite(ref.FE_SOURCE_GRAPHIC, "o", ref.COMPOSE_XOR, result="x")
                                     .composite("x", ref.FE_SOURCE_ALPHA, ref.COMPOSE_ATOP), swap),
        ("arithmetic", lambda f: f.blur(0.8, result="b").composite(ref.FE_SOURCE_GRAPHIC, "b", (0.2, 0.6, 0.5, 0.05)), swap.scale(2.0)),
        ("matrix_morph", lambda f: f.color_matrix(ref.FE_SOURCE_GRAPHIC, ref.COLOR_MATRIX_LUM, result="l")
                                    .morphology(0.8, 1.1, "max", "l", result="m").blend("m", ref.FE_SOURCE_GRAPHIC), swap.scale(2.0)),
        ("erode", lambda f: f.morphology(1.0, 1.0, "min", ref.FE_SOURCE_GRAPHIC), swap.scale(1.6)),
    ]
    import warnings
    for j, (name, build, tr) in enumerate(chains):
        flt = build(ref.Filter.empty())
        src = rand_layer((16, 19), 4, (6, 2), True, False)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            res = flt(tr, src)
        out[f"f{j}_in"], out[f"f{j}_tr"], out[f"f{j}_out"] = src.image, tr.m, res.image
        prims = []
        for ftype, attrs, inputs in flt.filters:
            a = []
            for v in attrs:
                if isinstance(v, np.ndarray):
                    out[f"f{j}_matrix{len(prims)}"] = v
                    a.append("matrix")
                elif isinstance(v, tuple):
                    a.append(list(v))
                else:
                    a.append(v)
            prims.append(dict(type=int(ftype), attrs=a, inputs=[int(i) for i in inputs]))
        meta["chain"].append(dict(name=name, prims=prims, in_offset=[int(v) for v in src.offset], out_offset=[int(v) for v in res.offset],
                                  out_pre_alpha=res.pre_alpha, out_linear_rgb=res.linear_rgb))

    # -- luminance mask ------------------------------------------------------------------------------------------------
    blob = ref.Path.from_svg("M20,30 C20,5 80,5 80,30 S110,85 60,90 C30,92 20,60 20,30 Z")
    ring = ref.Path.from_svg("M60,10 L75,95 L10,40 L110,40 L45,95 Z")
    for j, lin in enumerate([False, True]):
        target = ref.Scene.fill(blob, np.array([0.2, 0.5, 0.1, 0.9]))
        mask_scene = ref.Scene.group([ref.Scene.fill(ring, np.array([0.9, 0.9, 0.2, 1.0]), "evenodd"),
                                      ref.Scene.fill(blob, np.array([0.1, 0.3, 0.6, 0.7]))])
        scene = target.mask(mask_scene, False)
        layer, _hull = scene.render(swap, viewport=[0, 0, 120, 140], linear_rgb=lin)
        out[f"k{j}_out"] = layer.image
        meta["mask"].append(dict(linear_rgb=lin, offset=[int(v) for v in layer.offset], pre_alpha=layer.pre_alpha,
                                 out_linear_rgb=layer.linear_rgb))
    out["meta"] = np.array(json.dumps(meta))
    save("filter_kat.npz", **out)



def gen_svg(ref, fonts) -> None:
    """SVG loader known answers (svg_scene, S:2803-3083): the hand-written documents of tests/svg_cases.py loaded by the
    reference, stored as scene dumps."""
    import warnings
    sys.path.insert(0, os.path.dirname(HERE))
    from tests import svg_cases
    out, meta = {}, []
    for idx, (name, text, width) in enumerate(svg_cases.CASES):
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            scene, _ids, size = ref.svg_scene_from_str(text, width=width, fonts=ref.FontsDB())  # document fonts only
        m = dict(name=name, width=width, none=scene is None, size=None if size is None else [float(v) for v in size])
        if scene is not None:
            d = Dumper(ref)
            out[f"{idx}_tree"] = np.array(json.dumps(d.node(scene)))
            for k, v in d.arrays().items():
                out[f"{idx}_{k}"] = v
            m["unsupported"] = sorted(d.unsupported)
            if size is not None and "str" not in d.unsupported:
                # what the reference's CLI would draw (S:3836-3870): x/y swap, whole document viewport
                w, h = int(size[0]), int(size[1])
                with warnings.catch_warnings():
                    warnings.simplefilter("ignore")
                    res = scene.render(ref.Transform().matrix(0, 1, 0, 1, 0, 0), viewport=[0, 0, h, w], linear_rgb=False)
                if res is not None:
                    cl = res[0].convert(pre_alpha=True, linear_rgb=False)
                    canvas = np.zeros((h, w, 4))
                    ref.canvas_merge_at(canvas, cl.image, cl.offset)
                    out[f"{idx}_canvas"] = canvas
                    m["canvas"] = [h, w]
                    if name in ("gradients", "clip_mask_opacity", "filters", "patterns"):
                        # the same documents composited in linear RGB (the command line's --linear-rgb)
                        with warnings.catch_warnings():
                            warnings.simplefilter("ignore")
                            res_lin = scene.render(ref.Transform().matrix(0, 1, 0, 1, 0, 0), viewport=[0, 0, h, w], linear_rgb=True)
                        cl_lin = res_lin[0].convert(pre_alpha=True, linear_rgb=True)
                        canvas_lin = np.zeros((h, w, 4))
                        ref.canvas_merge_at(canvas_lin, cl_lin.image, cl_lin.offset)
                        out[f"{idx}_canvas_lin"] = canvas_lin.astype(np.float32)
                    # ... and the file it would write (S:3866-3877), without and with a background colour
                    page = ref.Layer(ref.canvas_merge_at(np.zeros((h, w, 4)), cl.image, cl.offset), (0, 0), pre_alpha=True,
                                     linear_rgb=False)
                    out[f"{idx}_png"] = np.frombuffer(page.write_png().getvalue(), dtype=np.uint8)
                    out[f"{idx}_png_bg"] = np.frombuffer(page.background(ref.svg_color("#fdf6e3")).write_png().getvalue(),
                                                         dtype=np.uint8)
        if name == "nested_svg_use":
            # the command line's -id / -t / -fg options (S:3836-3850): one element on its own bounding box, scaled, with a
            # default foreground for shapes without a fill
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                _scene, ids, _size = ref.svg_scene_from_str(text.replace(' fill="green"', ""), width=width, fonts=ref.FontsDB(),
                                                            fg=ref.svg_color("#b5651d"))
                tr = ref.Transform().matrix(0, 1, 0, 1, 0, 0) @ ref.svg_transform("scale(3) rotate(10)")
                layer, _hull = ids["leaf"].render(tr, linear_rgb=False)
            out["by_id_png"] = np.frombuffer(layer.write_png().getvalue(), dtype=np.uint8)
        meta.append(m)
    out["meta"] = np.array(json.dumps(meta))
    save("svg_kat.npz", **out)
    print("  svg:", [(m["name"], m.get("unsupported")) for m in meta])


def gen_svgfuzz(ref) -> None:
    """Grammar-generated documents (oracle/fuzz_svg_frontend.py) the reference can draw, with the canvas it draws: random
    nestings of groups, viewports, clips, masks, patterns, gradients, strokes -- end-to-end pins beyond the hand-written
    documents.  The document text itself is stored, so that the fixture does not depend on the generator."""
    import random
    import warnings
    import fuzz_svg_frontend as fuzz
    out, meta = {}, []
    # (SVGFUZZ_START / SVGFUZZ_COUNT / SVGFUZZ_OUT: ad-hoc sets for hunting, written outside tests/golden)
    seed = int(os.environ.get("SVGFUZZ_START", "0"))
    want_n = int(os.environ.get("SVGFUZZ_COUNT", "40"))
    last_seed = seed + 100 * want_n
    while len(meta) < want_n and seed < last_seed:
        r = random.Random(seed)
        seed += 1
        text = fuzz.document(r)
        width = r.choice([None, None, 77, 300])
        if os.environ.get("SVGFUZZ_WIDTH"):  # (hunting at larger sizes: more bands and tiles per shape)
            width = int(os.environ["SVGFUZZ_WIDTH"])
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            try:
                with fuzz.time_limit(60):
                    scene, _ids, size = ref.svg_scene_from_str(text, width=width, fonts=ref.FontsDB())
                    if scene is None or size is None:
                        continue
                    w, h = int(size[0]), int(size[1])
                    if w * h > int(os.environ.get("SVGFUZZ_MAXPX", str(400 * 400))) or w < 8 or h < 8:
                        continue
                    res = scene.render(ref.Transform().matrix(0, 1, 0, 1, 0, 0), viewport=[0, 0, h, w], linear_rgb=False)
                    if res is None:
                        continue
                    cl = res[0].convert(pre_alpha=True, linear_rgb=False)
                    canvas = np.zeros((h, w, 4))
                    ref.canvas_merge_at(canvas, cl.image, cl.offset)
            except (Exception, fuzz.TooSlow):  # noqa: BLE001  (a paint the reference cannot draw, an empty group, a stroke it never finishes, ...)
                continue
        if not np.isfinite(canvas).all() or canvas[..., 3].max() < 0.05:
            continue
        k = len(meta)
        out[f"{k}_canvas"] = canvas.astype(np.float32)
        if k % 3 == 0:  # every third document also composited in linear RGB
            try:
                with warnings.catch_warnings(), fuzz.time_limit(60):
                    warnings.simplefilter("ignore")
                    res = scene.render(ref.Transform().matrix(0, 1, 0, 1, 0, 0), viewport=[0, 0, h, w], linear_rgb=True)
                    cl = res[0].convert(pre_alpha=True, linear_rgb=True)
                    lin = np.zeros((h, w, 4))
                    ref.canvas_merge_at(lin, cl.image, cl.offset)
                    out[f"{k}_canvas_lin"] = lin.astype(np.float32)
            except (Exception, fuzz.TooSlow):  # noqa: BLE001
                pass
        crop = None
        if k % 3 == 1:  # every third document also through a window: the viewport cropping of S:966-975 on every leaf
            rr = random.Random(seed * 7919)
            rows, cols = rr.randrange(12, max(13, h // 2)), rr.randrange(12, max(13, w // 2))
            r0, c0 = rr.randrange(-6, h - rows + 6), rr.randrange(-6, w - cols + 6)
            try:
                with warnings.catch_warnings(), fuzz.time_limit(60):
                    warnings.simplefilter("ignore")
                    res = scene.render(ref.Transform().matrix(0, 1, 0, 1, 0, 0), viewport=[r0, c0, rows, cols], linear_rgb=False)
                    win = np.zeros((rows, cols, 4))
                    if res is not None:
                        cl = res[0].convert(pre_alpha=True, linear_rgb=False)
                        ref.canvas_merge_at(win, cl.image, (cl.offset[0] - r0, cl.offset[1] - c0))
                    out[f"{k}_canvas_crop"] = win.astype(np.float32)
                    crop = [r0, c0, rows, cols]
            except (Exception, fuzz.TooSlow):  # noqa: BLE001
                crop = None
        free = None
        if k % 3 == 2:  # every third document also without a viewport: the layer grows to whatever the content reaches
            try:
                with warnings.catch_warnings(), fuzz.time_limit(60):
                    warnings.simplefilter("ignore")
                    res = scene.render(ref.Transform().matrix(0, 1, 0, 1, 0, 0), linear_rgb=False)
                    if res is not None and res[0].image.shape[0] * res[0].image.shape[1] <= 600 * 600:
                        cl = res[0].convert(pre_alpha=True, linear_rgb=False)
                        out[f"{k}_layer_free"] = cl.image.astype(np.float32)
                        free = [int(cl.offset[0]), int(cl.offset[1])]
            except (Exception, fuzz.TooSlow):  # noqa: BLE001
                free = None
        meta.append(dict(seed=seed - 1, width=width, size=[h, w], text=text, crop=crop, free=free))
    out["meta"] = np.array(json.dumps(meta))
    if os.environ.get("SVGFUZZ_OUT"):
        np.savez_compressed(os.environ["SVGFUZZ_OUT"], **out)
    else:
        save("svg_fuzz_kat.npz", **out)
    print("  svg fuzz seeds:", [m["seed"] for m in meta])


HOSTUTIL_PATHS = [
    "M1,2 L3.5,4 H10 V-2.25 z",
    "M0,0 Q5,10 10,0 T20,0 C25,5 30,-5 35,0 S45,5 50,0",
    "M10,30 a10,6 30 1 0 20,5 z m 5,5 l 1e-3,2 z",
    "M0,0 h10 A5,5 0 0 1 20,10 L0,10 z M3,3 v4 h4 z M50,50 z",
    "M1234.5678,0.000123 L-1e6,1e-7 C1,2 3,4 5.55555555,6 Q1,1 2,2",
    "M.5.5L-1-1 1e1,2E+1 3.e0,4\n\tl+1,-.25e1z",
    "m1,1 2,2 3,3 m1,1 1,0z l5,5 t1,1 s2,2 3,3",
    "M0,0 A0,5 0 0 1 5,5 L9,9 a3,0 0 1 1 1,1 A2,2 45 1 1 9,9 z",
    "M5,5 z z M1,1",
    "",
]
HOSTUTIL_SCENES = ["basic_shapes", "groups_transforms_style", "clip_mask_opacity", "nested_svg_use", "text_svg_font"]


def gen_hostutil(ref) -> None:
    """Host-side conveniences: Path.to_svg (S:1204), Path.__repr__ (S:1436), Path.transform (S:1182), ConvexHull.path
    (S:2025), Scene.__repr__ (S:796), Scene.to_path (S:754) on the hand-written documents, Layer.background (S:166)."""
    import warnings
    sys.path.insert(0, os.path.dirname(HERE))
    from tests import svg_cases
    out, meta = {}, dict(paths=[], scenes=[])
    tr = ref.Transform().translate(3, -2).rotate(0.3).scale(1.5, 0.75)
    for idx, d in enumerate(HOSTUTIL_PATHS):
        path = ref.Path.from_svg(d)
        moved = path.transform(tr)
        types, params, sizes = pack_segments(ref, moved)
        out[f"p{idx}_types"], out[f"p{idx}_params"], out[f"p{idx}_sizes"] = types, params, sizes
        meta["paths"].append(dict(d=d, to_svg=path.to_svg(), repr=repr(path), moved_to_svg=moved.to_svg(), empty=path.is_empty()))
    meta["tr"] = [float(x) for x in tr.m[:2].ravel()]
    hull = ref.ConvexHull([[0, 0], [4, 0], [4, 3], [2, 1], [0, 3], [2, 5]])
    meta["hull"] = dict(points=[[float(x) for x in p] for p in hull.points], repr=repr(hull.path()))
    cases = {name: (text, width) for name, text, width in svg_cases.CASES}
    view = ref.Transform().matrix(0, 1, 0, 1, 0, 0).scale(0.5)
    for idx, name in enumerate(HOSTUTIL_SCENES):
        text, width = cases[name]
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            scene, _ids, _size = ref.svg_scene_from_str(text, width=width, fonts=ref.FontsDB())
        l, c = gather_defs(ref, scene.to_path(view))
        out[f"s{idx}_lines"], out[f"s{idx}_cubics"] = l, c
        meta["scenes"].append(dict(name=name, repr=repr(scene)))
    rng = np.random.default_rng(4242)
    a = rng.uniform(0, 1, (9, 7, 1))
    img = np.concatenate([rng.uniform(0, 1, (9, 7, 3)) * a, a], axis=-1)
    colour = np.array([0.1, 0.25, 0.05, 0.5])
    for idx, (pre, lin) in enumerate([(True, True), (False, False), (True, False)]):
        src = ref.Layer(img if pre else np.concatenate([img[..., :3] / a, a], axis=-1), (2, 5), pre_alpha=pre, linear_rgb=lin)
        out[f"bg{idx}_in"], out[f"bg{idx}_out"] = src.image, src.background(colour).image
    meta["bg"] = dict(colour=[float(x) for x in colour], flags=[[True, True], [False, False], [True, False]])
    out["meta"] = np.array(json.dumps(meta))
    save("hostutil_kat.npz", **out)


# --------------------------------------------------------------------------------------
# module-level canvas functions (S:235-416) called directly on arrays
# --------------------------------------------------------------------------------------
def gen_canvasfn(ref) -> None:
    from functools import partial

    rng = np.random.default_rng(0xCA17A5)
    out, meta = {}, []

    def image(r, c, ch):
        a = rng.uniform(0, 1, size=(r, c, 1))
        a[rng.uniform(size=a.shape) < 0.25] = 0.0
        a[rng.uniform(size=a.shape) < 0.25] = 1.0
        return a if ch == 1 else np.concatenate([rng.uniform(0, 1, size=(r, c, 3)) * a, a], axis=-1)

    modes = [ref.COMPOSE_OVER, ref.COMPOSE_OUT, ref.COMPOSE_IN, ref.COMPOSE_ATOP, ref.COMPOSE_XOR, (0.3, 0.5, 0.4, 0.05)]
    for mode in modes:
        for chd, chs in ((4, 4), (4, 1), (1, 4), (1, 1)):
            i = len(meta)
            d, s_ = image(7, 9, chd), image(7, 9, chs)
            out[f"{i}_dst"], out[f"{i}_src"] = d, s_
            out[f"{i}_out"] = ref.canvas_compose(mode, d, s_)
            meta.append(dict(fn="compose", mode=list(mode) if isinstance(mode, tuple) else int(mode)))
    for off in ((2, 3), (-3, -2), (9, 11), (40, 2), (-1, 10)):
        i = len(meta)
        base, over = image(12, 14, 4), image(6, 5, 4) * 1.5   # (> 1 so that the clip of the touched region shows)
        out[f"{i}_base"], out[f"{i}_over"] = base.copy(), over
        res = ref.canvas_merge_at(base, over, off)
        out[f"{i}_out"] = base
        meta.append(dict(fn="merge_at", offset=list(off), none=res is None))
    for mode, full in ((ref.COMPOSE_OVER, False), (ref.COMPOSE_OVER, True), (ref.COMPOSE_XOR, True), (ref.COMPOSE_IN, True)):
        i = len(meta)
        layers = [(image(int(rng.integers(3, 9)), int(rng.integers(3, 9)), 4), (int(rng.integers(-4, 6)), int(rng.integers(-4, 6))))
                  for _ in range(3)]
        res, roff = ref.canvas_merge_union(layers, full=full, blend=partial(ref.canvas_compose, mode))
        for j, (im, o) in enumerate(layers):
            out[f"{i}_in{j}"] = im
        out[f"{i}_out"] = res
        meta.append(dict(fn="union", mode=int(mode), full=full, offsets=[list(o) for _, o in layers], offset=[int(v) for v in roff]))
    for mode, ch0 in ((ref.COMPOSE_IN, 1), (ref.COMPOSE_OVER, 4), (ref.COMPOSE_IN, 4)):
        i = len(meta)
        layers = [(image(8, 9, ch0), (0, 1)), (image(7, 8, 4), (2, 0)), (image(9, 9, 4), (1, 2))]
        res, roff = ref.canvas_merge_intersect(layers, blend=partial(ref.canvas_compose, mode))
        for j, (im, o) in enumerate(layers):
            out[f"{i}_in{j}"] = im
        out[f"{i}_out"] = res
        meta.append(dict(fn="intersect", mode=int(mode), offsets=[list(o) for _, o in layers], offset=[int(v) for v in roff]))
    i = len(meta)
    disjoint = [(image(3, 3, 4), (0, 0)), (image(3, 3, 4), (10, 10))]
    meta.append(dict(fn="intersect_none", none=ref.canvas_merge_intersect(disjoint) is None))
    canvas, tr = ref.canvas_create(5, 3, bg=np.array([0.1, 0.2, 0.3, 1.0]))
    out["create_canvas"], out["create_m"] = canvas, np.asarray(tr.m, dtype=np.float64)
    out["meta"] = np.array(json.dumps(meta))
    save("canvasfn_kat.npz", **out)


def gen_gradlong(ref) -> None:
    """Gradients with more stops than the device code carries inline (the reference loops over any number, S:1671-1683)."""
    rng = np.random.default_rng(0x57095)
    out, meta = {}, []

    def stops(n):
        offs = np.sort(rng.uniform(0, 1, size=n))
        offs[0], offs[-1] = 0.0, 1.0
        offs[n // 2] = offs[n // 2 - 1]            # a repeated offset (a hard colour step)
        cols = []
        for _ in range(n):
            a = rng.uniform(0.2, 1.0)
            cols.append(np.concatenate([rng.uniform(0, 1, size=3) * a, [a]]))
        return [(float(o), c) for o, c in zip(offs, cols)]

    cases = [("linear", 33, "pad"), ("linear", 75, "reflect"), ("radial", 40, "repeat"), ("radial", 200, "pad")]
    for kind, n, spread in cases:
        st = stops(n)
        if kind == "linear":
            paint = ref.GradLinear(np.array([2.0, 3.0]), np.array([40.0, 25.0]), st, None, spread, False, None)
        else:
            paint = ref.GradRadial(np.array([20.0, 18.0]), 17.0, None, None, st, None, spread, False, None)
        i = len(meta)
        pts = rng.uniform(-10, 50, size=(400, 2))
        out[f"{i}_off"] = np.array([o for o, _ in st])
        out[f"{i}_rgba"] = np.array([c for _, c in st])
        out[f"{i}_pts"] = pts
        out[f"{i}_lin"] = paint.fill(pts, linear_rgb=True)
        out[f"{i}_srgb"] = paint.fill(pts, linear_rgb=False)
        path = ref.Path.from_svg("M3,2 H45 V38 H3 Z")
        layer, _ = path.fill(ref.Transform().matrix(0, 1, 0, 1, 0, 0), paint, linear_rgb=False)
        out[f"{i}_image"] = layer.image
        meta.append(dict(kind=kind, n=n, spread=spread, offset=[int(v) for v in layer.offset]))
    out["meta"] = np.array(json.dumps(meta))
    save("gradlong_kat.npz", **out)


# --------------------------------------------------------------------------------------
# S. the headline workload's generator, rendered by the reference itself
# --------------------------------------------------------------------------------------
def gen_synth(ref) -> None:
    """svgrasterize.py_amd/synth.py scenes drawn by the REFERENCE: every path through Path.fill (S:995-1019, linear RGB: the paint
    is taken as already in the compositing space), the fills through Layer.compose(OVER) (S:177-207) and canvas_merge_at onto a
    transparent canvas (S:304-327).  Pins oracle.render_solid -- the checker of the bench line -- and the GPU to the reference on
    the bench's own kind of drawing."""
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from svgrasterize_amd import synth

    swap = ref.Transform().matrix(0, 1, 0, 1, 0, 0)
    out = {}
    meta = []
    for size, n in ((256, 48), (700, 300)):
        sc = synth.make_scene(size, n)
        off = sc["path_seg_off"]
        layers = []
        for p in range(n):
            cub = sc["segs"][off[p]:off[p + 1]].reshape(-1, 4, 2)
            sub = [(ref.PATH_CUBIC, np.array(c, dtype=np.float64)) for c in cub]
            path = ref.Path([sub])
            res = path.fill(swap, sc["path_paint"][p].copy(), fill_rule="evenodd" if sc["path_rule"][p] else None,
                            viewport=[0, 0, size, size], linear_rgb=True)
            if res is not None:
                layers.append(res[0])
        top = ref.Layer.compose(layers, ref.COMPOSE_OVER, linear_rgb=True)
        canvas = np.zeros((size, size, 4))
        ref.canvas_merge_at(canvas, top.image, top.offset)
        key = f"s{size}_n{n}"
        out[key + "_canvas"] = canvas
        meta.append(dict(key=key, size=size, paths=n, layers=len(layers)))
    out["meta"] = np.array(json.dumps(meta))
    save("synth_kat.npz", **out)


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--full", action="store_true", help="also render the full-size configs (slow)")
    ap.add_argument("--only", default=None)
    args = ap.parse_args()
    os.makedirs(GOLD, exist_ok=True)
    ref = ref_loader.load()
    fonts = ref.FontsDB()
    fonts.register_file(ref.DEFAULT_FONTS)
    todo = lambda k: args.only is None or args.only == k
    if todo("coverage"):
        gen_coverage(ref)
    if todo("flatten"):
        tiger, _, _ = ref.svg_scene_from_filepath(os.path.join(DEMO, "icons/tiger.svg"), width=2048, fonts=fonts)
        gen_flatten(ref, tiger)
    if todo("stroke"):
        tiger, _, _ = ref.svg_scene_from_filepath(os.path.join(DEMO, "icons/tiger.svg"), width=2048, fonts=fonts)
        gen_stroke(ref, tiger)
    if todo("png"):
        gen_png(ref)
    if todo("filters"):
        gen_filters(ref)
    if todo("svg"):
        gen_svg(ref, fonts)
    if todo("svgfuzz"):
        gen_svgfuzz(ref)
    if todo("hostutil"):
        gen_hostutil(ref)
    if todo("synth"):
        gen_synth(ref)
    if todo("mask"):
        gen_mask(ref)
    if todo("compose"):
        gen_compose(ref)
    if todo("canvasfn"):
        gen_canvasfn(ref)
    if todo("gradlong"):
        gen_gradlong(ref)
    if todo("gradient"):
        gen_gradient(ref)
    if todo("tiger"):
        gen_scene(ref, fonts, "tiger", "icons/tiger.svg", 2048, [1 / 16, 1 / 8], args.full, crop=[900, 700, 96, 128])
    if todo("material"):
        gen_scene(ref, fonts, "material", "material-design.svg", 4096, [1 / 16], args.full, crop=[1000, 1000, 160, 192])
    if todo("icons"):
        gen_scene(ref, fonts, "icons", "icons.svg", None, [1.0], False, store_layer=False)
    if todo("icons4096"):
        # config 5 at its stated size (4096 x 1051): scene dump + sparse pins of the reference's full render (--full)
        gen_scene(ref, fonts, "icons4096", "icons.svg", 4096, [], args.full, store_layer=False)
    if todo("iconset"):
        # the reference's other demo icons at thumbnail size: real-world mixes of gradients, clips, masks, filters, strokes
        for svg in sorted(os.listdir(os.path.join(DEMO, "icons"))):
            if svg.endswith(".svg") and svg != "tiger.svg":
                gen_scene(ref, fonts, "icon_" + os.path.splitext(svg)[0].replace("-", "_"), "icons/" + svg, 192, [1.0], False,
                          store_layer=False)
    if todo("prompt"):
        gen_scene(ref, fonts, "prompt", "prompt.svg", 256, [1.0], False)


if __name__ == "__main__":
    main()
