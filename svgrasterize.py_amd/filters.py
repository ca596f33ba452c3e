"""Filter effects (reference S:1716-1944): feGaussianBlur (the one on the hot path) and the other
primitives the reference implements -- feOffset, feMerge, feBlend, feComposite, feColorMatrix, feMorphology (8f-4).

``Filter`` keeps the reference's (names, filters) structure so scene dumps replay unchanged.  The blur weights are
built on the host exactly like ``blur_kernel`` does (a few thousand numbers); every per-pixel operation runs on the
GPU (``svgr_layer_convolve``, ``svgr_layer_blend``, ``svgr_layer_color_matrix``, ``svgr_layer_morphology``)."""
from __future__ import annotations

import math
from typing import NamedTuple

import numpy as np

from .geometry import Transform
from .layer import Layer

import warnings

FE_BLEND, FE_COLOR_MATRIX, FE_COMPONENT_TRANSFER, FE_COMPOSITE, FE_CONVOLVE_MATRIX = 0, 1, 2, 3, 4  # S:1716-1730
FE_DIFFUSE_LIGHTING, FE_DISPLACEMENT_MAP, FE_FLOOD, FE_GAUSSIAN_BLUR, FE_MERGE = 5, 6, 7, 8, 9
FE_MORPHOLOGY, FE_OFFSET, FE_SPECULAR_LIGHTING, FE_TILE, FE_TURBULENCE = 10, 11, 12, 13, 14
COLOR_MATRIX_LUM = np.array([[0, 0, 0, 0, 0], [0, 0, 0, 0, 0], [0, 0, 0, 0, 0], [0.2125, 0.7154, 0.0721, 0, 0]], dtype=np.float64)
# feColorMatrix type="saturate" / "hueRotate" (the SVG 1.1 filter chapter's constants; S:1740-1747): the colour block is
# _HUE_BASE + cos(a) * _HUE_COS + sin(a) * _HUE_SIN; saturate(s) is the same with (cos, sin) := (s, 0)
_HUE_BASE = np.array([[0.213, 0.715, 0.072]] * 3, dtype=np.float64)
_HUE_COS = np.array([[0.787, -0.715, -0.072], [-0.213, 0.285, -0.072], [-0.213, -0.715, 0.928]], dtype=np.float64)
_HUE_SIN = np.array([[-0.213, -0.715, 0.928], [0.143, 0.140, -0.283], [-0.787, 0.715, 0.072]], dtype=np.float64)


def _hue_matrix(c: float, s: float) -> np.ndarray:
    matrix = np.eye(4, 5)
    matrix[:3, :3] = np.dot(np.stack([_HUE_BASE, _HUE_COS, _HUE_SIN]).T, [1, c, s]).T
    return matrix


def color_matrix_hue_rotate(angle: float) -> np.ndarray:
    """4x5 colour matrix of a hue rotation by ``angle`` radians (S:1947-1951)."""
    return _hue_matrix(math.cos(angle), math.sin(angle))


def color_matrix_saturate(value: float) -> np.ndarray:
    """4x5 colour matrix of feColorMatrix ``saturate`` (S:1954-1957)."""
    return _hue_matrix(value, 0)


FE_SOURCE_ALPHA = "SourceAlpha"
FE_SOURCE_GRAPHIC = "SourceGraphic"


_KERNEL_MEMO: dict = {}   # (matrix bytes, sigma) -> weights: a document's blurs repeat from render to render (and among its nodes)


def blur_kernel(transform: Transform, sigma):
    """Gaussian weights on the pixel grid for a blur given in user space (S:1903-1944); None = no-op.

    The values are the reference's own numpy expressions (its fixtures pin them bit for bit: numpy's vectorised ``exp``
    and its pairwise ``sum`` have no counterpart in libm), evaluated once per (matrix, sigma): the 37 blurs of icons.svg
    cost a dictionary look-up each on every render after the first."""
    memo_key = (transform.key(), float(sigma[0]), float(sigma[1]))
    hit = _KERNEL_MEMO.get(memo_key, _KERNEL_MEMO)
    if hit is not _KERNEL_MEMO:
        return hit
    if len(_KERNEL_MEMO) > 1024:
        _KERNEL_MEMO.clear()
    weights = _KERNEL_MEMO[memo_key] = _blur_weights(transform, float(sigma[0]), float(sigma[1]))
    if weights is not None:
        weights.setflags(write=False)   # (shared between the renders that look it up)
    return weights


def _blur_weights(transform: Transform, sigma_x: float, sigma_y: float):
    origin = transform([0, 0])
    # device pixels per user unit along the two axes; a deviation below half a pixel is raised to half a pixel unless BOTH are
    # (then the blur is the identity)
    per_x, per_y = np.linalg.norm(transform(np.eye(2)) - origin, axis=1)
    small_x, small_y = per_x * sigma_x < 0.5, per_y * sigma_y < 0.5
    if small_x and small_y:
        return None
    if small_x:
        sigma_x = 0.5 / per_x
    elif small_y:
        sigma_y = 0.5 / per_y
    # the kernel's footprint: 2.5 deviations to either side, mapped to the pixel grid, made odd in both directions
    reach_x, reach_y = 2.5 * sigma_x, 2.5 * sigma_y
    frame = transform([[-reach_x, -reach_y], [-reach_x, reach_y], [reach_x, reach_y], [reach_x, -reach_y]]) - origin
    low, high = frame.min(axis=0).astype(int), frame.max(axis=0).astype(int)
    kw, kh = (int(n) + (~int(n) & 1) for n in high - low)
    # pixel centres of the footprint, taken back to user space without the translation
    back = transform.invert
    ix, iy = np.indices((kw, kh)).astype(np.float64)
    centres = np.concatenate([ix[..., None], iy[..., None]], axis=2) + [-kw / 2 + 0.5, -kh / 2 + 0.5]
    user = back(centres)
    user -= back([0, 0])
    weights = np.exp(-np.square(user) / (2 * np.square(np.array([sigma_x, sigma_y])))).prod(axis=-1)
    return weights / weights.sum()


class Filter(NamedTuple):
    names: dict
    filters: list  # [(type, attrs, inputs)]

    @classmethod
    def empty(cls) -> "Filter":
        return cls({FE_SOURCE_ALPHA: 0, FE_SOURCE_GRAPHIC: 1}, [])

    def add_filter(self, type, attrs, inputs, result=None) -> "Filter":
        names, filters = dict(self.names), list(self.filters)
        args = []
        for name in inputs:
            idx = None if name is None else self.names.get(name)
            args.append(len(filters) + 1 if idx is None else idx)  # default: previous result
        if result is not None:
            names[result] = len(filters) + 2
        filters.append((type, attrs, args))
        return Filter(names, filters)

    def offset(self, dx, dy, input=None, result=None) -> "Filter":
        return self.add_filter(FE_OFFSET, (dx, dy), [input], result)

    def merge(self, inputs, result=None) -> "Filter":
        return self.add_filter(FE_MERGE, tuple(), inputs, result)

    def blur(self, std_x, std_y=None, input=None, result=None) -> "Filter":
        return self.add_filter(FE_GAUSSIAN_BLUR, (std_x, std_y), [input], result)

    def blend(self, in1, in2, mode=None, result=None) -> "Filter":
        return self.add_filter(FE_BLEND, (mode,), [in1, in2], result)

    def composite(self, in1, in2, mode=None, result=None) -> "Filter":
        return self.add_filter(FE_COMPOSITE, (mode,), [in1, in2], result)

    def color_matrix(self, input, matrix, result=None) -> "Filter":
        return self.add_filter(FE_COLOR_MATRIX, (matrix,), [input], result)

    def morphology(self, rx, ry, method, input, result=None) -> "Filter":
        return self.add_filter(FE_MORPHOLOGY, (rx, ry, method), [input], result)

    def __call__(self, transform: Transform, source: Layer) -> Layer:
        """Execute the filter chain on `source` (S:1801-1831)."""
        stack: list = [None, source.convert(pre_alpha=False, linear_rgb=True)]

        def get(i):
            if i == 0 and stack[0] is None:  # SourceAlpha, built only when referenced
                alpha = source.image[..., -1:] * np.array([0, 0, 0, 1])
                stack[0] = Layer(alpha, source.offset, pre_alpha=True, linear_rgb=True)
            return stack[i]

        for ftype, attrs, inputs in self.filters:
            args = [get(i) for i in inputs]
            if ftype == FE_GAUSSIAN_BLUR:
                std_x, std_y = attrs
                kernel = blur_kernel(transform, (std_x, std_x if std_y is None else std_y))
                res = args[0] if kernel is None else args[0].convolve(kernel)
            elif ftype == FE_OFFSET:  # S:1844-1850
                dx, dy = attrs
                x, y = args[0].offset
                tx, ty = transform(transform.invert([x, y]) + [dx, dy])
                res = args[0].translate(int(tx) - x, int(ty) - y)
            elif ftype == FE_MERGE:  # S:1867-1871
                res = Layer.compose(args, linear_rgb=True)
            elif ftype == FE_BLEND:  # S:1874-1879 (the reference composes OVER whatever the mode)
                warnings.warn("feBlend is not properly supported")
                res = Layer.compose([args[1], args[0]], linear_rgb=True)
            elif ftype == FE_COMPOSITE:  # S:1882-1886
                res = Layer.compose([args[1], args[0]], attrs[0], linear_rgb=True)
            elif ftype == FE_COLOR_MATRIX:  # S:1834-1841
                matrix = attrs[0]
                if not isinstance(matrix, np.ndarray) or matrix.shape != (4, 5):
                    warnings.warn(f"invalid color matrix: {matrix}")
                    res = args[0]
                else:
                    res = args[0].color_matrix(matrix)
            elif ftype == FE_MORPHOLOGY:  # S:1853-1864
                rx, ry, method = attrs
                ux, uy = transform([[rx, 0], [0, ry]]) - transform([[0, 0], [0, 0]])
                x, y = int(np.linalg.norm(ux) * 2), int(np.linalg.norm(uy) * 2)
                res = args[0] if x < 1 or y < 1 else args[0].morphology(x, y, method)
            else:
                raise ValueError(f"unsupported filter type: {ftype}")
            stack.append(res)
        return get(len(stack) - 1)
