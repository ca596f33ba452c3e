"""Paint servers of the hot path (reference S:1544-1712): linear and radial gradients, patterns.

The per-pixel work (pixel centre -> user space -> gradient offset -> spread -> stop interpolation ->
times mask) runs in the HIP kernel ``k_gradient_fill``; what the reference computes ONCE per fill with
numpy (inverse transforms, ``vec``, the focal-circle scalars, the colour-space conversion of the stops)
is computed here with the same numpy expressions and handed over through ``svgr_gradient``."""
from __future__ import annotations

import ctypes as C
from typing import NamedTuple

import numpy as np

from . import _abi

_SPREAD = {"pad": 0, "repeat": 1, "reflect": 2}


_STOPS_MEMO: dict = {}  # (id(stops), linear_rgb) -> (stops, converted): a gradient is usually filled many times
_STOP_ARRAYS: dict = {}  # ... -> (stops, offsets array, colours array): what the ABI struct points at
_D6, _D2 = C.c_double * 6, C.c_double * 2


def _stops_colorspace(stops, linear_rgb: bool):
    """grad_stops_colorspace, S:1686-1695: premultiplied linear RGBA stops -> target colour space."""
    from .geometry import solid_paint

    key = (id(stops), bool(linear_rgb))
    hit = _STOPS_MEMO.get(key)
    if hit is not None and hit[0] is stops:
        return hit[1]
    out = [(float(o), solid_paint(np.asarray(c, dtype=np.float64), linear_rgb)) for o, c in stops]
    if len(_STOPS_MEMO) > 4096:
        _STOPS_MEMO.clear()
    _STOPS_MEMO[key] = (stops, out)
    return out


class _GradMixin:
    def _common(self, g: "_abi.Gradient", user_tr, linear_rgb: bool):
        if self.spread not in _SPREAD:
            raise ValueError(f"invalid spread method: {self.spread}")
        g.spread = _SPREAD[self.spread]
        g.user_m6 = _D6(*np.asarray(user_tr.m, dtype=np.float64)[:2].ravel().tolist())
        if self.transform is not None:
            g.has_gt = 1
            g.gt_m6 = _D6(*np.asarray(self.transform.invert.m, dtype=np.float64)[:2].ravel().tolist())
        key = (id(self.stops), bool(linear_rgb))
        hit = _STOP_ARRAYS.get(key)
        if hit is not None and hit[0] is self.stops:
            _stops, off, col = hit
        else:
            stops = _stops_colorspace(self.stops, linear_rgb)
            if not stops:
                raise ValueError("a gradient needs at least one stop")
            off = np.ascontiguousarray([o for o, _ in stops], dtype=np.float64)
            col = np.ascontiguousarray([c for _, c in stops], dtype=np.float64).reshape(-1, 4)
            if len(_STOP_ARRAYS) > 4096:
                _STOP_ARRAYS.clear()
            _STOP_ARRAYS[key] = (self.stops, off, col)
        g.n_stops = len(off)
        g.stop_off = _abi.ptr(off)
        g.stop_rgba = _abi.ptr(col)
        return off, col  # keep alive until the call returns


    def fill(self, pixels, linear_rgb: bool = True) -> np.ndarray:
        """The gradient's colour at each coordinate of ``pixels`` (..., 2), user space: an array (..., 4) of premultiplied
        RGBA in the requested colour space (GradLinear.fill S:1553-1563, GradRadial.fill S:1577-1651).  Evaluated by the
        same device code ``Path.fill`` uses on the pixel grid, here on the caller's points."""
        from .geometry import Transform

        pts = np.ascontiguousarray(pixels, dtype=np.float64)
        if pts.shape[-1:] != (2,):
            raise ValueError("pixels must be an array of (x, y) coordinates")
        n = pts.size // 2
        if n == 0:
            return np.zeros(pts.shape[:-1] + (4,))
        g, keep = self.abi(Transform(), linear_rgb)
        ctx = _abi.Context.get()
        src = ctx.alloc(n * 16)
        src.upload(pts)
        out = ctx.alloc(n * 32)
        _abi._check(ctx.lib.svgr_gradient_eval(ctx.handle, C.byref(g), src.handle, n, out.handle))
        del keep
        return out.download(pts.shape[:-1] + (4,), np.float64)


class GradLinear(_GradMixin, NamedTuple("GradLinear", [("p0", object), ("p1", object), ("stops", list), ("transform", object),
                                                        ("spread", str), ("bbox_units", bool), ("linear_rgb", object)])):
    def abi(self, user_tr, linear_rgb: bool):
        g = _abi.Gradient()
        g.kind = 1
        keep = self._common(g, user_tr, linear_rgb)
        p0 = np.asarray(self.p0, dtype=np.float64)
        vec = np.asarray(self.p1, dtype=np.float64) - p0  # S:1561
        g.p0 = _D2(*p0.tolist())
        g.vec = _D2(*vec.tolist())
        g.vv = float(np.dot(vec, vec))
        return g, keep


class GradRadial(_GradMixin, NamedTuple("GradRadial", [("center", object), ("radius", float), ("fcenter", object),
                                                        ("fradius", object), ("stops", list), ("transform", object),
                                                        ("spread", str), ("bbox_units", bool), ("linear_rgb", object)])):
    def abi(self, user_tr, linear_rgb: bool):
        g = _abi.Gradient()
        keep = self._common(g, user_tr, linear_rgb)
        center = np.asarray(self.center, dtype=np.float64)
        g.center = _D2(*center.tolist())
        g.radius = float(self.radius)
        if self.fcenter is None and self.fradius is None:  # S:1605
            g.kind = 2
            return g, keep
        g.kind = 3
        fcenter = center if self.fcenter is None else np.asarray(self.fcenter, dtype=np.float64)
        fradius = self.fradius or 0
        cd = center - fcenter                      # S:1619
        rd = self.radius - fradius                 # S:1621
        g.fcenter = _D2(*fcenter.tolist())
        g.fradius = float(fradius)
        g.cd = _D2(*cd.tolist())
        g.rd = float(rd)
        g.a = float((cd ** 2).sum() - rd ** 2)     # S:1622
        g.frad_rd = float(fradius * rd)            # S:1623
        g.frad2 = float(fradius ** 2)              # S:1624
        g.excl_enabled = int(fradius != self.radius)
        g.excl_thresh = float(fradius / (fradius - self.radius)) if fradius != self.radius else 0.0  # S:1644
        return g, keep


def is_gradient(paint) -> bool:
    return isinstance(paint, (GradLinear, GradRadial))


def needs_mask(paint) -> bool:
    """Paints that are applied to the path's coverage mask (``Path.mask`` first), i.e. everything but a solid colour."""
    return isinstance(paint, (GradLinear, GradRadial, Pattern))


def gradient_fill(paint, mask_layer, hull, transform, linear_rgb: bool):
    """Path.fill, gradient branch (S:1021-1047): returns the RGBA Layer = gradient * mask."""
    from .layer import Layer

    if paint.bbox_units:
        user_tr = hull.bbox_transform(transform).invert
    else:
        user_tr = transform.invert
    if paint.linear_rgb is not None:
        linear_rgb = paint.linear_rgb
    g, keep = paint.abi(user_tr, linear_rgb)
    ctx = _abi.Context.get()
    rows, cols = mask_layer.height, mask_layer.width
    out = ctx.alloc(rows * cols * 32)
    mbuf = mask_layer._device()
    bbox = (C.c_int64 * 4)(int(mask_layer.x), int(mask_layer.y), rows, cols)
    _abi._check(ctx.lib.svgr_gradient_fill(ctx.handle, C.byref(g), mbuf.handle, bbox, out.handle))
    del keep
    return Layer._from_device(out, (rows, cols, 4), mask_layer.offset, pre_alpha=True, linear_rgb=linear_rgb)


class Pattern(NamedTuple):
    """A scene repeated over the plane as a paint (S:1698-1710).  ``x, y, width, height`` is the cell in pattern space,
    ``transform`` the patternTransform, ``bbox_units`` whether the cell is in objectBoundingBox units; the content is in
    ``scene_view_box`` coordinates if that is set, else in objectBoundingBox units if ``scene_bbox_units``."""

    scene: object
    scene_bbox_units: bool
    scene_view_box: object
    x: float
    y: float
    width: float
    height: float
    transform: object
    bbox_units: bool

    def bbox(self):
        return (self.x, self.y, self.width, self.height)


def pattern_fill(paint: Pattern, mask_layer, hull, transform, linear_rgb: bool):
    """Path.fill, pattern branch (S:1049-1094): the RGBA Layer = repeated tile * mask, or None for an empty tile.

    The tile is the pattern's scene rendered once (recursively, on the device) under the fill's transform without its
    translation.  Every pixel centre of the mask is taken to pattern space, reduced modulo the cell, taken back and
    truncated to an integer offset into the tile; ``k_pattern_fill`` does that per pixel with the reference's
    operations and multiplies by the coverage.  What is computed once per fill stays here, in the reference's numpy
    expressions."""
    from .layer import Layer
    from .svg import viewbox_transform  # noqa: PLC0415 (svg imports this module)

    pat_tr = transform.no_translate()
    if paint.scene_view_box:
        if paint.bbox_units:
            px, py, pw, ph = paint.bbox()
            _hx, _hy, hw, hh = hull.bbox(transform)
            box = (px * hw, py * hh, pw * hw, ph * hh)
        else:
            box = paint.bbox()
        pat_tr = pat_tr @ viewbox_transform(box, paint.scene_view_box)
    elif paint.scene_bbox_units:
        pat_tr = hull.bbox_transform(pat_tr)
    pat_tr = pat_tr @ paint.transform
    result = paint.scene.render(pat_tr, linear_rgb=linear_rgb)
    if result is None:
        return None
    tile, _tile_hull = result

    repeat_tr = transform
    if paint.bbox_units:
        repeat_tr = hull.bbox_transform(repeat_tr)
    repeat_tr = (repeat_tr @ paint.transform).no_translate()
    corners = repeat_tr(np.array([[0, 0], [paint.width, 0], [0, paint.height], [paint.width, paint.height]], dtype=np.float64))
    max_x, max_y = corners.max(axis=0).astype(int)
    min_x, min_y = corners.min(axis=0).astype(int)

    pat = _abi.PatternArgs()
    pat.inv_m6 = (C.c_double * 6)(*np.asarray(repeat_tr.invert.m, dtype=np.float64)[:2].ravel())
    pat.fwd_m6 = (C.c_double * 6)(*np.asarray(repeat_tr.m, dtype=np.float64)[:2].ravel())
    pat.cell = (C.c_double * 4)(paint.x, paint.y, paint.width, paint.height)
    pat.min_xy = (C.c_int64 * 2)(int(min_x), int(min_y))
    pat.pat_shape = (C.c_int64 * 2)(int(max_x - min_x) + 1, int(max_y - min_y) + 1)
    pat.tile_bbox = (C.c_int64 * 4)(int(tile.x) - int(min_x), int(tile.y) - int(min_y), tile.height, tile.width)
    ctx = _abi.Context.get()
    rows, cols = mask_layer.height, mask_layer.width
    out = ctx.alloc(rows * cols * 32)
    bbox = (C.c_int64 * 4)(int(mask_layer.x), int(mask_layer.y), rows, cols)
    # (bound to names: a host-resident layer's _device() is a temporary buffer that must outlive the call)
    tile_buf, mask_buf = tile._device(), mask_layer._device()
    _abi._check(ctx.lib.svgr_pattern_fill(ctx.handle, C.byref(pat), tile_buf.handle, mask_buf.handle, bbox, out.handle))
    return Layer._from_device(out, (rows, cols, 4), mask_layer.offset, pre_alpha=tile.pre_alpha, linear_rgb=tile.linear_rgb)
