"""ctypes front-end of the CPU oracle (oracle/svgr_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py.  The product package never imports this module.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libsvgr_oracle.so")

_f64p = np.ctypeslib.ndpointer(dtype=np.float64, flags="C_CONTIGUOUS")
_i64p = np.ctypeslib.ndpointer(dtype=np.int64, flags="C_CONTIGUOUS")
_u8p = np.ctypeslib.ndpointer(dtype=np.uint8, flags="C_CONTIGUOUS")
_lib = None


def build(force: bool = False) -> str:
    src = os.path.join(HERE, "svgr_oracle.c")
    if force or not os.path.exists(LIB_PATH) or os.path.getmtime(LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", HERE, "-B", "libsvgr_oracle.so"], stdout=subprocess.DEVNULL)
    return LIB_PATH


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        build()
    L = C.CDLL(LIB_PATH)
    L.orc_transform_points.argtypes = [_f64p, _f64p, C.c_int64, _f64p]
    L.orc_matmul3.argtypes = [_f64p, _f64p, _f64p]
    L.orc_flatness.argtypes = [_f64p, C.c_int64, _f64p]
    L.orc_split.argtypes = [_f64p, C.c_int64, _f64p]
    L.orc_flatten.argtypes = [_f64p, C.c_int64, C.c_double, C.c_void_p, C.c_int64]
    L.orc_flatten.restype = C.c_int64
    L.orc_bbox.argtypes = [_f64p, C.c_int64, C.c_void_p, _i64p]
    L.orc_bbox.restype = C.c_int
    L.orc_line_coverage.argtypes = [_f64p, C.c_int64, C.c_int64, _f64p]
    L.orc_mask.argtypes = [_f64p, C.c_int64, _i64p, C.c_int64, C.c_int64, C.c_int, _f64p]
    for name in ("orc_pre_to_straight", "orc_straight_to_pre", "orc_linear_to_srgb", "orc_srgb_to_linear"):
        getattr(L, name).argtypes = [_f64p, C.c_int64]
    L.orc_paint_for_fill.argtypes = [_f64p, C.c_int, _f64p]
    L.orc_fill_solid.argtypes = [_f64p, C.c_int64, _f64p, _f64p]
    L.orc_compose_over.argtypes = [_f64p, _i64p, C.c_int64, C.c_int64, _f64p, _i64p, C.c_int64, C.c_int64, C.c_int, C.c_int]
    L.orc_compose_in.argtypes = [_f64p, _i64p, C.c_int64, C.c_int64, _f64p, _i64p, C.c_int64, C.c_int64, C.c_int]
    L.orc_render_solid.argtypes = [_f64p, _u8p, _i64p, C.c_int64, _u8p, _f64p, _i64p, C.c_int, _f64p, C.c_void_p]
    L.orc_render_solid.restype = C.c_int
    L.orc_render_solid_strips.argtypes = [_f64p, _u8p, _i64p, C.c_int64, _u8p, _f64p, _i64p, C.c_int, _f64p, C.c_void_p,
                                          C.c_int, C.c_int]
    L.orc_render_solid_strips.restype = C.c_int
    L.orc_visible_count.argtypes = [C.c_int]
    L.orc_visible_count.restype = None
    L.orc_visible_get.argtypes = []
    L.orc_visible_get.restype = C.c_int64
    _lib = L
    return L


def _c(a, dtype=np.float64):
    return np.ascontiguousarray(a, dtype=dtype)


RULES = {None: 0, "nonzero": 0, "evenodd": 1}


def transform_points(m, pts):
    """m: 3x3 (or 2x3) affine; pts (..., 2)."""
    m = np.asarray(m, dtype=np.float64)
    m6 = _c(m[:2].ravel())
    pts = _c(pts)
    out = np.empty_like(pts)
    lib().orc_transform_points(m6, pts.reshape(-1), pts.size // 2, out.reshape(-1))
    return out


def matmul3(a, b):
    out = np.empty((3, 3))
    lib().orc_matmul3(_c(a), _c(b), out)
    return out


def flatness(cubics):
    cubics = _c(cubics).reshape(-1, 4, 2)
    out = np.empty(len(cubics))
    lib().orc_flatness(cubics.reshape(-1), len(cubics), out)
    return out


def split(cubics):
    cubics = _c(cubics).reshape(-1, 4, 2)
    out = np.empty((2 * len(cubics), 4, 2))
    lib().orc_split(cubics.reshape(-1), len(cubics), out.reshape(-1))
    return out


def flatten(cubics, tol=0.1):
    cubics = _c(cubics).reshape(-1, 4, 2)
    n = lib().orc_flatten(cubics.reshape(-1), len(cubics), tol, None, 0)
    if n < 0:
        raise RuntimeError(f"orc_flatten failed: {n}")
    out = np.empty((n, 2, 2))
    if n:
        lib().orc_flatten(cubics.reshape(-1), len(cubics), tol, out.ctypes.data_as(C.c_void_p), n)
    return out


def bbox(edges, viewport=None):
    edges = _c(edges).reshape(-1, 2, 2)
    out = np.zeros(4, dtype=np.int64)
    vp = None
    if viewport is not None:
        vp_arr = _c(viewport, np.int64)
        vp = vp_arr.ctypes.data_as(C.c_void_p)
    ok = lib().orc_bbox(edges.reshape(-1), len(edges), vp, out)
    return (tuple(int(v) for v in out) if ok else None)


def line_coverage(trace, line):
    lib().orc_line_coverage(trace, trace.shape[0], trace.shape[1], _c(line).reshape(-1))
    return trace


def mask(edges, bb, rule=None):
    edges = _c(edges).reshape(-1, 2, 2)
    r0, c0, rows, cols = bb
    out = np.empty((rows, cols))
    lib().orc_mask(edges.reshape(-1), len(edges), np.array([r0, c0], dtype=np.int64), rows, cols, RULES[rule], out.reshape(-1))
    return out


def path_edges(lines, cubics, m=None, tol=0.1):
    """Edges of one path in presentation space, reference order: explicit lines then flattened cubics."""
    lines = _c(lines).reshape(-1, 2, 2)
    cubics = _c(cubics).reshape(-1, 4, 2)
    if m is not None:
        lines = transform_points(m, lines)
        cubics = transform_points(m, cubics)
    return np.concatenate([lines, flatten(cubics, tol)])


def path_mask(lines, cubics, m=None, rule=None, viewport=None):
    """Path.mask restated: returns (mask(rows, cols), (r0, c0)) or None."""
    edges = path_edges(lines, cubics, m)
    if len(edges) == 0:
        return None
    bb = bbox(edges, viewport)
    if bb is None:
        return None
    return mask(edges, bb, rule), (bb[0], bb[1]), edges


def paint_for_fill(paint, linear_rgb):
    out = np.empty(4)
    lib().orc_paint_for_fill(_c(paint), int(bool(linear_rgb)), out)
    return out


def fill_solid(mask_img, paint):
    mask_img = _c(mask_img)
    out = np.empty(mask_img.shape[:2] + (4,))
    lib().orc_fill_solid(mask_img.reshape(-1), mask_img.shape[0] * mask_img.shape[1], _c(paint), out.reshape(-1))
    return out


def convert(img, pre_alpha, linear_rgb, to_pre_alpha, to_linear_rgb):
    """Layer.convert restated for 4-channel images (S:129-164); returns a new array."""
    L = lib()
    out = _c(img).copy()
    n = out.size // 4
    flat = out.reshape(-1)
    if linear_rgb != to_linear_rgb:
        if pre_alpha:
            L.orc_pre_to_straight(flat, n)
            pre_alpha = False
        (L.orc_srgb_to_linear if to_linear_rgb else L.orc_linear_to_srgb)(flat, n)
    if pre_alpha != to_pre_alpha:
        (L.orc_straight_to_pre if to_pre_alpha else L.orc_pre_to_straight)(flat, n)
    return out


def compose_over(layers):
    """canvas_merge_union(full=False) + OVER restated. layers = [(image(r,c,ch), (r0,c0))]."""
    r0 = min(o[0] for _, o in layers)
    c0 = min(o[1] for _, o in layers)
    r1 = max(o[0] + im.shape[0] for im, o in layers)
    c1 = max(o[1] + im.shape[1] for im, o in layers)
    out = np.zeros((r1 - r0, c1 - c0, 4))
    doff = np.array([r0, c0], dtype=np.int64)
    for i, (im, off) in enumerate(layers):
        im = _c(im)
        lib().orc_compose_over(out.reshape(-1), doff, out.shape[0], out.shape[1], im.reshape(-1),
                               np.array(off, dtype=np.int64), im.shape[0], im.shape[1], im.shape[2], int(i == 0))
    return out, (r0, c0)


def compose_in(layers):
    """canvas_merge_intersect + IN restated; returns None on empty intersection."""
    r0 = max(o[0] for _, o in layers)
    c0 = max(o[1] for _, o in layers)
    r1 = min(o[0] + im.shape[0] for im, o in layers)
    c1 = min(o[1] + im.shape[1] for im, o in layers)
    if r0 >= r1 or c0 >= c1:
        return None
    first, foff = layers[0]
    out = first[r0 - foff[0]: r1 - foff[0], c0 - foff[1]: c1 - foff[1]]
    if out.shape[2] == 1:
        out = np.broadcast_to(out, out.shape[:2] + (4,))
    out = np.ascontiguousarray(out, dtype=np.float64).copy()
    doff = np.array([r0, c0], dtype=np.int64)
    for im, off in layers[1:]:
        im = _c(im)
        lib().orc_compose_in(out.reshape(-1), doff, out.shape[0], out.shape[1], im.reshape(-1),
                             np.array(off, dtype=np.int64), im.shape[0], im.shape[1], im.shape[2])
    return out, (r0, c0)


def render_solid(segs, seg_kind, path_seg_off, path_rule, path_paint, viewport, clip01=True, strips=1, threads=1):
    """Whole solid-fill scene on the CPU. Returns (canvas f64 (rows, cols, 4), P, E).
    strips > 1: the canvas as that many row strips, each an independent render through the reference's own viewport
    cropping (S:968-971), on `threads` OpenMP threads (E is not counted then: 0)."""
    vp = _c(viewport, np.int64)
    canvas = np.zeros((int(vp[2]), int(vp[3]), 4))
    stats = np.zeros(2, dtype=np.int64)
    args = (_c(segs).reshape(-1), _c(seg_kind, np.uint8), _c(path_seg_off, np.int64), len(path_seg_off) - 1,
            _c(path_rule, np.uint8), _c(path_paint).reshape(-1), vp, int(clip01), canvas.reshape(-1),
            stats.ctypes.data_as(C.c_void_p))
    if strips > 1:
        rc = lib().orc_render_solid_strips(*args, int(strips), int(max(1, threads)))
    else:
        rc = lib().orc_render_solid(*args)
    if rc != 0:
        raise RuntimeError(f"orc_render_solid failed: {rc}")
    return canvas, int(stats[0]), int(stats[1])


def host_threads(limit=16):
    """This process's share of the host cores, capped (the strips of a render are independent: one thread each)."""
    import os

    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    return max(1, min(limit, n))


# --------------------------------------------------------------------------------------
# gradients and blur (config 5), restated in numpy.  S:n = reference line n.
# --------------------------------------------------------------------------------------
def _affine(m, pts):
    """Transform.__call__ (S:531-534) on an (..., 2) array: numpy's own matmul supplies the fma form."""
    m = np.asarray(m, dtype=np.float64)
    return pts @ m[:2, :2].T + m[:2, 2]


def gradient_image(kind, bbox, user_m, gt_m, spread, stop_off, stop_rgba, p0=None, p1=None, center=None, radius=None,
                   fcenter=None, fradius=None):
    """paint.fill(user_tr(grad_pixels(bbox))) of S:1021-1031 for stops already in the target colour space."""
    r0, c0, rows, cols = bbox
    xs, ys = np.indices((rows, cols)).astype(np.float64)
    px = np.concatenate([xs[..., None], ys[..., None]], axis=2) + [r0 + 0.5, c0 + 0.5]  # grad_pixels S:1653-1658
    px = _affine(user_m, px)
    if gt_m is not None:
        px = _affine(gt_m, px)  # the caller passes the INVERSE gradientTransform (S:1559 / S:1603)
    mask = None
    if kind == "linear":
        vec = np.asarray(p1, float) - np.asarray(p0, float)
        offset = (px - p0) @ vec / np.dot(vec, vec)
    elif fcenter is None and fradius is None:
        offset = (px - center) / radius
        offset = np.sqrt((offset * offset).sum(axis=-1))
    else:
        fc = np.asarray(center if fcenter is None else fcenter, float)
        fr = fradius or 0
        cd = np.asarray(center, float) - fc
        pd = px - fc
        rd = radius - fr
        a = (cd ** 2).sum() - rd ** 2
        b = (pd * cd).sum(axis=-1) + fr * rd
        c = (pd ** 2).sum(axis=-1) - fr ** 2
        det = b * b - a * c
        if (det < 0).any():
            mask = det >= 0
        with np.errstate(invalid="ignore"):
            t0 = np.sqrt(det)
        off_all = np.maximum((b + t0) / a, (b - t0) / a)
        if mask is None:
            offset = off_all
        else:
            offset = np.where(mask, off_all, 0.0)
            if fr != radius:
                mask = mask & (offset > (fr / (fr - radius)))
    if spread == "repeat":
        offset = np.modf(offset)[0]
    elif spread == "reflect":
        offset = np.fabs(np.remainder(offset + 1.0, 2.0) - 1.0)
    elif spread != "pad":
        raise ValueError(f"invalid spread method: {spread}")
    out = np.zeros(offset.shape + (4,))
    out[offset <= stop_off[0]] = stop_rgba[0]
    out[offset > stop_off[-1]] = stop_rgba[-1]
    for s in range(len(stop_off) - 1):
        o0, o1 = stop_off[s], stop_off[s + 1]
        sel = np.logical_and(offset > o0, offset <= o1)
        ratio = ((offset[sel] - o0) / (o1 - o0))[..., None]
        out[sel] += (1 - ratio) * stop_rgba[s] + ratio * stop_rgba[s + 1]
    if mask is not None:
        out[~mask] = 0.0
    return out


def convolve_full(image, kernel):
    """Layer.convolve (S:106-118): full 2-D convolution, direct summation (the reference lets scipy pick
    FFT, which differs by ~1e-16)."""
    image = np.asarray(image, dtype=np.float64)
    kw, kh = kernel.shape
    out = np.zeros((image.shape[0] + kw - 1, image.shape[1] + kh - 1, image.shape[2]))
    for i in range(kw):
        for j in range(kh):
            out[i: i + image.shape[0], j: j + image.shape[1]] += image * kernel[i, j]
    return out
