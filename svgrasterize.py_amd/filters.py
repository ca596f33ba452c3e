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


def blur_kernel(transform: Transform, sigma):
    """Gaussian weights on the pixel grid for a blur given in user space (S:1903-1944); None = no-op."""
    sigma_x, sigma_y = sigma
    scale_x, scale_y = np.linalg.norm(transform(np.eye(2)) - transform([0, 0]), axis=1)
    if scale_x * sigma_x < 0.5 and scale_y * sigma_y < 0.5:
        return None  # below half a pixel in both directions
    elif scale_x * sigma_x < 0.5:
        sigma_x = 0.5 / scale_x
    elif scale_y * sigma_y < 0.5:
        sigma_y = 0.5 / scale_y
    sig = np.array([sigma_x, sigma_y])
    ext = 2.5  # support in sigmas
    corners = [[-ext * sigma_x, -ext * sigma_y], [-ext * sigma_x, ext * sigma_y],
               [ext * sigma_x, ext * sigma_y], [ext * sigma_x, -ext * sigma_y]]
    box = transform(corners) - transform([0, 0])
    lo_x, lo_y = box.min(axis=0).astype(int)
    hi_x, hi_y = box.max(axis=0).astype(int)
    kw, kh = hi_x - lo_x, hi_y - lo_y
    kw += ~kw & 1  # odd sizes
    kh += ~kh & 1
    inv = transform.invert
    xs, ys = np.indices((kw, kh)).astype(np.float64)
    grid = np.concatenate([xs[..., None], ys[..., None]], axis=2) + [-kw / 2 + 0.5, -kh / 2 + 0.5]  # pixel centres
    pts = inv(grid)
    pts -= inv([0, 0])  # drop the translation
    weights = np.exp(-np.square(pts) / (2 * np.square(sig))).prod(axis=-1)
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
