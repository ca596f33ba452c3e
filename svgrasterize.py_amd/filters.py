"""feGaussianBlur (reference S:1750-1831, 1890-1944): the only filter primitive on the hot path.

``Filter`` keeps the reference's (names, filters) structure so scene dumps replay unchanged; of the
primitive types only FE_GAUSSIAN_BLUR is executed (the others are outside the accelerated path,
SURVEY 8f-4).  The kernel weights are built on the host exactly like ``blur_kernel`` does (a few
thousand numbers); the convolution itself runs on the GPU (``svgr_layer_convolve``)."""
from __future__ import annotations

from typing import NamedTuple

import numpy as np

from .geometry import Transform
from .layer import Layer

FE_GAUSSIAN_BLUR = 8
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

    def blur(self, std_x, std_y=None, input=None, result=None) -> "Filter":
        return self.add_filter(FE_GAUSSIAN_BLUR, (std_x, std_y), [input], result)

    def __call__(self, transform: Transform, source: Layer) -> Layer:
        """Execute the filter chain on `source` (S:1801-1831)."""
        stack: list = [None, source.convert(pre_alpha=False, linear_rgb=True)]

        def get(i):
            if i == 0 and stack[0] is None:  # SourceAlpha, built only when referenced
                alpha = source.image[..., -1:] * np.array([0, 0, 0, 1])
                stack[0] = Layer(alpha, source.offset, pre_alpha=True, linear_rgb=True)
            return stack[i]

        for ftype, attrs, inputs in self.filters:
            if ftype != FE_GAUSSIAN_BLUR:
                raise NotImplementedError(f"filter primitive {ftype} is outside the accelerated path (SURVEY 8f-4)")
            std_x, std_y = attrs
            std_y = std_x if std_y is None else std_y
            layer = get(inputs[0])
            kernel = blur_kernel(transform, (std_x, std_y))
            stack.append(layer if kernel is None else layer.convolve(kernel))
        return get(len(stack) - 1)
