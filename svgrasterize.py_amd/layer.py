"""Layer: the value type of the hot path (reference S:61-233), with the image resident in HBM.

The reference's Layer is ``NamedTuple(image, offset, pre_alpha, linear_rgb)`` whose image is a
float64 numpy array of shape (rows, cols, 1|4).  This class keeps the same four fields, the same
tuple unpacking and the same methods, but the pixels live in a device buffer until somebody
reads ``.image``; from then on the materialised numpy array is the source of truth (callers such
as the reference's font_speciment.py mutate it in place) and device ops re-upload it.

compose / convert / opacity run as HIP kernels on double images with the reference's operation
order (csrc/svgr_hip.hip: k_layer_*).
"""
from __future__ import annotations

from typing import Sequence

import warnings

import ctypes as C

import numpy as np

from . import _abi

COMPOSE_OVER, COMPOSE_OUT, COMPOSE_IN, COMPOSE_ATOP, COMPOSE_XOR = 0, 1, 2, 3, 4
COMPOSE_PRE_ALPHA = {COMPOSE_OVER, COMPOSE_OUT, COMPOSE_IN, COMPOSE_ATOP, COMPOSE_XOR}
FLOAT = np.float64


def _bbox_arr(offset, shape):
    return (C.c_int64 * 4)(int(offset[0]), int(offset[1]), int(shape[0]), int(shape[1]))


class Layer:
    # `_ops`: the svgr_layer_convert ops the pixels in `_dev` still need to be what `pre_alpha` / `linear_rgb` say they are.  A
    # conversion of a device-resident layer is first only this note: the kernel that reads the layer next (compose OVER, opacity)
    # converts on the way in, everybody else gets the converted pixels from `_device()` / `.image` (one svgr_layer_convert_to).
    __slots__ = ["_host", "_dev", "_shape", "offset", "pre_alpha", "linear_rgb", "_ops"]

    def __init__(self, image, offset, pre_alpha: bool, linear_rgb: bool):
        image = np.asarray(image)
        if image.ndim != 3 or image.shape[2] not in (1, 4):
            raise ValueError("Layer image must have shape (rows, cols, 1|4)")
        self._host = image
        self._dev = None
        self._shape = tuple(image.shape)
        self.offset = offset
        self.pre_alpha = pre_alpha
        self.linear_rgb = linear_rgb
        self._ops = 0

    @classmethod
    def _from_device(cls, buf: "_abi.DeviceBuffer", shape, offset, pre_alpha, linear_rgb, ops: int = 0) -> "Layer":
        self = object.__new__(cls)
        self._host = None
        self._dev = buf
        self._shape = tuple(map(int, shape))
        self.offset = offset
        self.pre_alpha = pre_alpha
        self.linear_rgb = linear_rgb
        self._ops = ops
        return self

    # -- NamedTuple compatibility ---------------------------------------------------------
    def __iter__(self):
        yield self.image
        yield self.offset
        yield self.pre_alpha
        yield self.linear_rgb

    def __len__(self):
        return 4

    def __getitem__(self, i):
        return (self.image, self.offset, self.pre_alpha, self.linear_rgb)[i]

    @property
    def image(self) -> np.ndarray:
        if self._host is None:
            self._host = self._device().download(self._shape, FLOAT)
            self._dev = None  # the host array may be mutated by the caller from now on
        return self._host

    # -- device residency -----------------------------------------------------------------
    def _device(self) -> "_abi.DeviceBuffer":
        """Device buffer holding the current pixels as float64 (uploads a host-resident image)."""
        if self._host is not None:
            return _abi.Context.get().from_host(np.ascontiguousarray(self._host, dtype=FLOAT))
        if self._ops:   # (a noted conversion nobody folded into a kernel of their own: now)
            ctx = _abi.Context.get()
            buf = ctx.alloc(int(np.prod(self._shape)) * 8)
            _abi._check(ctx.lib.svgr_layer_convert_to(ctx.handle, buf.handle, self._dev.handle, self._shape[0] * self._shape[1], self._ops))
            self._dev, self._ops = buf, 0
        return self._dev

    @property
    def on_device(self) -> bool:
        return self._host is None

    # -- reference properties --------------------------------------------------------------
    @property
    def x(self) -> int:
        return self.offset[0]

    @property
    def y(self) -> int:
        return self.offset[1]

    @property
    def width(self) -> int:
        return self._shape[1]

    @property
    def height(self) -> int:
        return self._shape[0]

    @property
    def channels(self) -> int:
        return self._shape[2]

    @property
    def bbox(self):
        return (*self.offset, *self._shape[:2])

    def translate(self, x: int, y: int) -> "Layer":
        out = object.__new__(Layer)
        out._host, out._dev, out._shape, out._ops = self._host, self._dev, self._shape, self._ops
        out.offset = (self.x + x, self.y + y)
        out.pre_alpha, out.linear_rgb = self.pre_alpha, self.linear_rgb
        return out

    def _retag(self, pre_alpha, linear_rgb) -> "Layer":
        out = object.__new__(Layer)
        out._host, out._dev, out._shape, out._ops = self._host, self._dev, self._shape, self._ops
        out.offset, out.pre_alpha, out.linear_rgb = self.offset, pre_alpha, linear_rgb
        return out

    def _copy_device(self) -> "_abi.DeviceBuffer":
        ctx = _abi.Context.get()
        if self._host is not None:
            return ctx.from_host(np.ascontiguousarray(self._host, dtype=FLOAT))
        n = int(np.prod(self._shape))
        out = ctx.alloc(n * 8)
        if self._ops:   # (the copy is the conversion's output)
            _abi._check(ctx.lib.svgr_layer_convert_to(ctx.handle, out.handle, self._dev.handle, self._shape[0] * self._shape[1], self._ops))
        else:
            _abi._check(ctx.lib.svgr_buf_copy(ctx.handle, out.handle, self._dev.handle, n * 8))
        return out

    # -- Layer.convert  S:129-164 ----------------------------------------------------------
    def convert(self, pre_alpha: bool | None = None, linear_rgb: bool | None = None) -> "Layer":
        pre_alpha = self.pre_alpha if pre_alpha is None else pre_alpha
        linear_rgb = self.linear_rgb if linear_rgb is None else linear_rgb
        if self.channels == 1:
            return self._retag(pre_alpha, linear_rgb)  # single channel is alpha: flags only
        ops = 0
        cur_pre = self.pre_alpha
        if self.linear_rgb != linear_rgb:
            if cur_pre:
                ops |= _abi.CONVERT_PRE_TO_STRAIGHT
                cur_pre = False
            ops |= _abi.CONVERT_SRGB_TO_LINEAR if linear_rgb else _abi.CONVERT_LINEAR_TO_SRGB
        if cur_pre != pre_alpha:
            # the kernel applies 1,2,4,8 in that order; "straight -> pre" is last, "pre -> straight" first
            if pre_alpha:
                ops |= _abi.CONVERT_STRAIGHT_TO_PRE
            else:
                ops |= _abi.CONVERT_PRE_TO_STRAIGHT
        if ops == 0:
            return self
        ctx = _abi.Context.get()
        if self._host is not None:
            buf = self._copy_device()
            _abi._check(ctx.lib.svgr_layer_convert(ctx.handle, buf.handle, self._shape[0] * self._shape[1], ops))
            return Layer._from_device(buf, self._shape, self.offset, pre_alpha, linear_rgb)
        # device-resident: noted, not run (`_ops`).  On top of an earlier note the two sets run as one pass when the kernel's order
        # (1, 2, 4, 8) is theirs: every new op behind every noted one; otherwise the noted ones run now.
        if self._ops and (self._ops.bit_length() > (ops & -ops).bit_length() - 1):
            self._device()
        return Layer._from_device(self._dev, self._shape, self.offset, pre_alpha, linear_rgb, self._ops | ops)

    # -- Layer.background  S:166-169 -------------------------------------------------------
    def background(self, color) -> "Layer":
        """The layer over a solid background colour (premultiplied linear RGBA, like every paint)."""
        layer = self.convert(pre_alpha=True, linear_rgb=True)
        ctx = _abi.Context.get()
        buf = layer._copy_device()
        rgba = np.ascontiguousarray(color, dtype=np.float64).reshape(4)
        _abi._check(ctx.lib.svgr_layer_background(ctx.handle, buf.handle, layer._shape[0] * layer._shape[1], rgba.ctypes.data))
        return Layer._from_device(buf, layer._shape, layer.offset, True, True)

    def show(self, format=None) -> None:
        """Print the layer on the terminal through the optional ``imshow`` module (debugging aid, S:220-232)."""
        try:
            from imshow import show  # noqa: PLC0415
        except ImportError:
            warnings.warn("to be able to show layer on terminal imshow is required")
            return
        show(self.convert(pre_alpha=False, linear_rgb=False).image, format=format)
        print()

    # -- Layer.opacity  S:171-175 ----------------------------------------------------------
    def opacity(self, opacity: float, linear_rgb: bool = False) -> "Layer":
        layer = self.convert(pre_alpha=True, linear_rgb=linear_rgb)
        ctx = _abi.Context.get()
        if layer._host is not None:
            buf = layer._copy_device()
            _abi._check(ctx.lib.svgr_layer_scale(ctx.handle, buf.handle, int(np.prod(layer._shape)), float(opacity)))
        elif layer._ops:   # (the conversion and the factor in one pass)
            buf = ctx.alloc(int(np.prod(layer._shape)) * 8)
            _abi._check(ctx.lib.svgr_layer_convert_scale_to(ctx.handle, buf.handle, layer._dev.handle, layer._shape[0] * layer._shape[1],
                                                            layer._ops, float(opacity)))
        else:
            buf = ctx.alloc(int(np.prod(layer._shape)) * 8)
            _abi._check(ctx.lib.svgr_layer_scale_to(ctx.handle, buf.handle, layer._dev.handle, int(np.prod(layer._shape)), float(opacity)))
        return Layer._from_device(buf, layer._shape, layer.offset, True, linear_rgb)

    # -- Layer.convolve  S:106-118 ---------------------------------------------------------
    def color_matrix(self, matrix: np.ndarray) -> "Layer":
        """Apply a 4x5 colour matrix on straight-alpha linear RGBA (Layer.color_matrix, S:95-104)."""
        if not isinstance(matrix, np.ndarray) or matrix.shape != (4, 5):
            raise ValueError("expected 4x5 matrix")
        if self.channels != 4:
            raise ValueError("color_matrix expects an RGBA layer")
        layer = self.convert(pre_alpha=False, linear_rgb=True)
        ctx = _abi.Context.get()
        buf = layer._copy_device() if layer is self else layer._device()  # (a converted layer owns a fresh buffer)
        m = np.ascontiguousarray(matrix, dtype=FLOAT)
        _abi._check(ctx.lib.svgr_layer_color_matrix(ctx.handle, buf.handle, layer.height * layer.width, m.ctypes.data_as(_abi._P)))
        return Layer._from_device(buf, layer._shape, layer.offset, pre_alpha=False, linear_rgb=True)

    def morphology(self, x: int, y: int, method: str) -> "Layer":
        """Morphology = min / max pooling with stride 1 (Layer.morphology S:120-127, pooling S:419-468); the window is
        x rows by y columns of this layer's array, the result shrinks by the window and keeps the offset."""
        if method not in ("min", "max"):
            raise ValueError(f"invalid poll method: {method}")
        if self.channels != 4:
            raise ValueError("morphology expects an RGBA layer")
        layer = self.convert(pre_alpha=True, linear_rgb=True)
        rows, cols = layer.height, layer.width
        x, y = int(x), int(y)
        if x < 1 or y < 1 or x > rows + 1 or y > cols + 1:
            raise ValueError("morphology window does not fit the layer")  # (the reference: negative dimensions, S:455-461)
        ctx = _abi.Context.get()
        if x == rows + 1 or y == cols + 1:
            # One more than the layer: the reference's strided view has a zero-length axis and the result is an empty image
            # (S:455-466), i.e. the shape is eroded / dilated away.  Layers here hold at least one pixel: a transparent one.
            shape = (max(rows - x + 1, 1), max(cols - y + 1, 1), 4)
            out = ctx.alloc(shape[0] * shape[1] * 32)
            out.zero()
            return Layer._from_device(out, shape, layer.offset, pre_alpha=True, linear_rgb=True)
        shape = (rows - x + 1, cols - y + 1, 4)
        out = ctx.alloc(shape[0] * shape[1] * 32)
        src = layer._device()
        _abi._check(ctx.lib.svgr_layer_morphology(ctx.handle, out.handle, src.handle, rows, cols, x, y, int(method == "max")))
        return Layer._from_device(out, shape, layer.offset, pre_alpha=True, linear_rgb=True)

    def luminance_mask(self, linear_rgb: bool) -> "Layer":
        """RENDER_MASK's mask layer (S:733-736): luminance of the straight-alpha colour times alpha, one channel."""
        if self.channels != 4:
            raise ValueError("luminance mask expects an RGBA layer")
        layer = self.convert(pre_alpha=False, linear_rgb=linear_rgb)
        ctx = _abi.Context.get()
        n = layer.height * layer.width
        out = ctx.alloc(max(n, 1) * 8)
        src = layer._device()
        _abi._check(ctx.lib.svgr_layer_luminance(ctx.handle, out.handle, src.handle, n))
        return Layer._from_device(out, (layer.height, layer.width, 1), layer.offset, pre_alpha=False, linear_rgb=linear_rgb)

    def convolve(self, kernel: np.ndarray) -> "Layer":
        """Full 2-D convolution on straight-alpha linear RGBA; offset moves by half the kernel."""
        if self.channels != 4:
            raise ValueError("convolve expects an RGBA layer")
        layer = self.convert(pre_alpha=False, linear_rgb=True)
        kernel = np.ascontiguousarray(kernel, dtype=FLOAT)
        kw, kh = kernel.shape
        ctx = _abi.Context.get()
        rows, cols = layer.height, layer.width
        out_shape = (rows + kw - 1, cols + kh - 1, 4)
        out = ctx.alloc(out_shape[0] * out_shape[1] * 32)
        if layer._host is None and layer._ops:   # (the conversion rides in the blur's first pass)
            _abi._check(ctx.lib.svgr_layer_convolve_ops(ctx.handle, out.handle, layer._dev.handle, rows, cols,
                                                        _abi.ptr(kernel), kw, kh, layer._ops))
        else:
            src = layer._device()
            _abi._check(ctx.lib.svgr_layer_convolve(ctx.handle, out.handle, src.handle, rows, cols, _abi.ptr(kernel), kw, kh))
        offset = (int(layer.x - kw / 2), int(layer.y - kh / 2))
        return Layer._from_device(out, out_shape, offset, pre_alpha=False, linear_rgb=True)

    # -- Layer.compose  S:177-207 ----------------------------------------------------------
    @staticmethod
    def compose(layers: Sequence["Layer"], method: int = COMPOSE_OVER, linear_rgb: bool = False) -> "Layer | None":
        if not layers:
            return None
        if len(layers) == 1:
            return layers[0]  # returned as is, unconverted (S:190-191)
        arithmetic = isinstance(method, tuple) and len(method) == 4
        if method not in COMPOSE_PRE_ALPHA and not arithmetic:
            raise ValueError(f"invalid compose mode: {method}")
        if arithmetic or method not in (COMPOSE_OVER, COMPOSE_IN):
            # OUT / ATOP / XOR / feComposite arithmetic: every layer zero-extended to the union canvas, blended one
            # after the other (canvas_merge_union(full=True), S:348-361).  Arithmetic composes straight alpha (S:193).
            pre = not arithmetic
            conv = [l.convert(pre_alpha=pre, linear_rgb=linear_rgb) for l in layers]
            ctx = _abi.Context.get()
            r0 = min(int(l.x) for l in conv)
            c0 = min(int(l.y) for l in conv)
            r1 = max(int(l.x) + l.height for l in conv)
            c1 = max(int(l.y) + l.width for l in conv)
            shape = (r1 - r0, c1 - c0, 4)
            out = ctx.alloc(shape[0] * shape[1] * 32)
            out.zero()
            obb = _bbox_arr((r0, c0), shape)
            k4 = np.array(method if arithmetic else (0, 0, 0, 0), dtype=FLOAT)
            for i, l in enumerate(conv):
                src = l._device()
                if i == 0:
                    _abi._check(ctx.lib.svgr_layer_over(ctx.handle, out.handle, obb, src.handle, _bbox_arr(l.offset, l._shape),
                                                        l.channels, 1))
                else:
                    _abi._check(ctx.lib.svgr_layer_blend(ctx.handle, out.handle, obb, src.handle, _bbox_arr(l.offset, l._shape),
                                                         l.channels, 5 if arithmetic else int(method), k4.ctypes.data_as(_abi._P)))
            offset = (min(l.x for l in conv), min(l.y for l in conv))
            return Layer._from_device(out, shape, offset, pre, linear_rgb)
        conv = [l.convert(pre_alpha=True, linear_rgb=linear_rgb) for l in layers]
        ctx = _abi.Context.get()
        lib = ctx.lib
        if method == COMPOSE_OVER:
            r0 = min(int(l.x) for l in conv)
            c0 = min(int(l.y) for l in conv)
            r1 = max(int(l.x) + l.height for l in conv)
            c1 = max(int(l.y) + l.width for l in conv)
            shape = (r1 - r0, c1 - c0, 4)
            out = ctx.alloc(shape[0] * shape[1] * 32)
            # one pass over the union: every layer read once, converted as it is read (svgr_layer_compose_over)
            n = len(conv)
            srcs = [l._dev if l._host is None else l._device() for l in conv]   # (uploads of host-resident layers stay referenced)
            handles = (_abi._P * n)(*[b.handle for b in srcs])
            bbs = (C.c_int64 * (4 * n))(*[int(v) for l in conv for v in (l.offset[0], l.offset[1], l._shape[0], l._shape[1])])
            chs = (C.c_int32 * n)(*[l.channels for l in conv])
            ops = (C.c_uint32 * n)(*[l._ops if l._host is None else 0 for l in conv])
            _abi._check(lib.svgr_layer_compose_over(ctx.handle, out.handle, _bbox_arr((r0, c0), shape), n, handles, bbs, chs, ops))
            offset = (min(l.x for l in conv), min(l.y for l in conv))
            return Layer._from_device(out, shape, offset, True, linear_rgb)
        # COMPOSE_IN: intersection, S:382-416
        r0 = max(int(l.x) for l in conv)
        c0 = max(int(l.y) for l in conv)
        r1 = min(int(l.x) + l.height for l in conv)
        c1 = min(int(l.y) + l.width for l in conv)
        if r0 >= r1 or c0 >= c1:
            return None
        shape = (r1 - r0, c1 - c0, 4)
        out = ctx.alloc(shape[0] * shape[1] * 32)
        # one pass over the intersection: the first layer cropped, the others multiplied in, each converted as it is read
        n = len(conv)
        srcs = [l._dev if l._host is None else l._device() for l in conv]   # (uploads of host-resident layers stay referenced)
        handles = (_abi._P * n)(*[b.handle for b in srcs])
        bbs = (C.c_int64 * (4 * n))(*[int(v) for l in conv for v in (l.offset[0], l.offset[1], l._shape[0], l._shape[1])])
        chs = (C.c_int32 * n)(*[l.channels for l in conv])
        ops = (C.c_uint32 * n)(*[l._ops if l._host is None else 0 for l in conv])
        _abi._check(lib.svgr_layer_compose_in(ctx.handle, out.handle, _bbox_arr((r0, c0), shape), n, handles, bbs, chs, ops))
        offset = (max(l.x for l in conv), max(l.y for l in conv))
        return Layer._from_device(out, shape, offset, True, linear_rgb)

    def on_canvas(self, rows: int, cols: int) -> "Layer":
        """This layer merged onto a transparent (rows, cols) canvas at the origin and clipped to [0, 1]: the
        ``canvas_merge_at`` step in front of the PNG writer (S:304-327, S:3866-3870).  Stays on the device."""
        ctx = _abi.Context.get()
        layer = self.convert(pre_alpha=True)
        canvas = ctx.alloc(rows * cols * 32)
        canvas.zero()
        src = layer._device()  # (a host-resident layer's upload is a temporary: it must outlive the call)
        _abi._check(ctx.lib.svgr_layer_over(ctx.handle, canvas.handle, _bbox_arr((0, 0), (rows, cols)), src.handle,
                                            _bbox_arr(layer.offset, layer._shape), layer.channels, 0))
        _abi._check(ctx.lib.svgr_layer_clip01(ctx.handle, canvas.handle, rows * cols * 4))
        return Layer._from_device(canvas, (rows, cols, 4), (0, 0), True, layer.linear_rgb)

    def to_canvas_f32(self, rows: int, cols: int, clip01: bool = True) -> np.ndarray:
        """Place this layer on a zero (rows, cols, 4) canvas (canvas_merge_at, S:304-327) and
        return float32 premultiplied RGBA."""
        ctx = _abi.Context.get()
        layer = self.convert(pre_alpha=True)
        canvas = ctx.alloc(rows * cols * 32)
        canvas.zero()
        src = layer._device()
        _abi._check(ctx.lib.svgr_layer_over(ctx.handle, canvas.handle, _bbox_arr((0, 0), (rows, cols)),
                                            src.handle, _bbox_arr(layer.offset, layer._shape),
                                            layer.channels, 0))
        out32 = ctx.alloc(rows * cols * 16)
        _abi._check(ctx.lib.svgr_layer_to_f32(ctx.handle, out32.handle, canvas.handle, rows * cols * 4, int(clip01)))
        return out32.download((rows, cols, 4), np.float32)

    def to_rgba8(self) -> np.ndarray:
        """(rows, cols, 4) uint8: straight-alpha sRGB, ``np.round(image * 255).astype(np.uint8)`` (S:211-212, S:262),
        converted and quantised on the device; only the bytes cross PCIe (4 B per pixel instead of 32)."""
        if self.channels != 4:
            raise ValueError("Only RGBA layers are supported")
        ctx = _abi.Context.get()
        layer = self.convert(pre_alpha=False, linear_rgb=False)
        src = layer._device()
        rows, cols = layer._shape[0], layer._shape[1]
        out = ctx.alloc(max(rows * cols * 4, 4))
        _abi._check(ctx.lib.svgr_layer_to_rgba8(ctx.handle, out.handle, src.handle, rows * cols))
        return out.download((rows, cols, 4), np.uint8)

    def write_png(self, output=None, level: int = 9, threads: int = 1):
        """PNG of the layer (Layer.write_png, S:209-213): 8-bit RGBA, filter 0, one IDAT -- byte-identical to the
        reference's file at the reference's zlib level 9 (``level`` trades size for speed, the pixels are the same)."""
        return canvas_to_png(self.to_rgba8(), output, level=level, threads=threads)

    def __repr__(self):
        return "Layer(x={}, y={}, w={}, h={}, pre_alpha={}, linear_rgb={})".format(
            self.x, self.y, self.width, self.height, self.pre_alpha, self.linear_rgb
        )


def _deflate_parallel(raw: bytes, level: int, threads: int) -> bytes:
    """A zlib stream of ``raw`` built from independently compressed pieces, one per worker thread at a time (the pigz
    construction: every piece but the last is a raw deflate stream ended with a sync flush, so the concatenation is one
    valid deflate stream; zlib header in front, Adler-32 of the whole input behind).  zlib releases the GIL, so the
    pieces really run side by side.  Decodes to the same bytes as the one-shot stream; only the file bytes differ."""
    import zlib
    from concurrent.futures import ThreadPoolExecutor

    piece = max(1 << 20, -(-len(raw) // (threads * 4)))
    spans = [(o, min(o + piece, len(raw))) for o in range(0, len(raw), piece)] or [(0, 0)]
    view = memoryview(raw)

    def work(k):
        lo, hi = spans[k]
        comp = zlib.compressobj(level, zlib.DEFLATED, -15)
        return comp.compress(view[lo:hi]) + comp.flush(zlib.Z_FINISH if k == len(spans) - 1 else zlib.Z_SYNC_FLUSH)

    with ThreadPoolExecutor(max_workers=threads) as pool:
        parts = list(pool.map(work, range(len(spans))))
    # CMF/FLG: deflate, 32 KiB window; FLEVEL is informative only (2 bits), FCHECK makes the pair a multiple of 31
    cmf, flevel = 0x78, (0 if level < 2 else 1 if level < 6 else 2 if level == 6 else 3) << 6
    flg = flevel + (31 - ((cmf << 8) + flevel) % 31) % 31
    return bytes([cmf, flg]) + b"".join(parts) + (zlib.adler32(raw) & 0xFFFFFFFF).to_bytes(4, "big")


def canvas_to_png(canvas, output=None, level: int = 9, threads: int = 1):
    """(height, width, 4) -> PNG (canvas_to_png, S:249-274).  ``canvas`` is either the uint8 array of
    ``Layer.to_rgba8`` or float RGBA in [0, 1] (quantised like the reference: ``np.round(canvas * 255)``).
    ``threads`` > 1 compresses the scanlines in parallel pieces: same pixels, a different (slightly larger) file than the
    reference's, written several times faster (4096 x 4096 at level 9: seconds -> a few tenths)."""
    import io
    import struct
    import zlib

    canvas = np.asarray(canvas)
    if canvas.dtype != np.uint8:
        canvas = np.round(canvas * 255.0).astype(np.uint8)
    if canvas.ndim != 3 or canvas.shape[2] != 4:
        raise ValueError("Only RGBA layers are supported")
    height, width, _ = canvas.shape

    def pack(out, tag: bytes, data: bytes) -> None:
        out.write(struct.pack("!I", len(data)))
        out.write(tag)
        out.write(data)
        out.write(struct.pack("!I", 0xFFFFFFFF & zlib.crc32(data, zlib.crc32(tag))))

    # filter byte 0 in front of every scanline, then ONE deflate stream (deflate output does not depend on how the
    # input is chunked, so one call gives the bytes of the reference's row-by-row loop)
    rows = np.zeros((height, 1 + width * 4), dtype=np.uint8)
    rows[:, 1:] = canvas.reshape(height, width * 4)
    if threads > 1:
        data = _deflate_parallel(rows.tobytes(), level, threads)
    else:
        comp = zlib.compressobj(level=level)
        data = comp.compress(rows.tobytes()) + comp.flush()
    output = io.BytesIO() if output is None else output
    output.write(b"\x89PNG\r\n\x1a\n")
    pack(output, b"IHDR", struct.pack("!2I5B", width, height, 8, 6, 0, 0, 0))
    pack(output, b"IDAT", data)
    pack(output, b"IEND", b"")
    return output


# ---------------------------------------------------------------------------------------------------------------------
# Module-level canvas functions of the reference (S:235-416) for callers that use them directly on numpy arrays.  The
# arithmetic runs in the same device kernels as Layer.compose (k_layer_over / k_layer_in / k_layer_blend); the arrays
# cross PCIe once in each direction per call, so inside a render the Layer methods (which stay in HBM) are the fast way.
# ---------------------------------------------------------------------------------------------------------------------
def canvas_create(width, height, bg=None):
    """(canvas, transform): a float64 (height, width, 4) canvas, transparent or filled with ``bg``, and the transform from
    (x, y) to its pixel coordinates (canvas_create, S:235-246)."""
    from .geometry import Transform  # noqa: PLC0415

    if bg is None:
        canvas = np.zeros((height, width, 4), dtype=FLOAT)
    else:
        canvas = np.array(np.broadcast_to(bg, (height, width, 4)), dtype=FLOAT)
    return canvas, Transform().matrix(0, 1, 0, 1, 0, 0)


def _image3(image, what):
    a = np.asarray(image, dtype=FLOAT)
    if a.ndim == 2:
        a = a[..., None]
    if a.ndim != 3 or a.shape[2] not in (1, 4):
        raise ValueError(f"{what}: expected an image of shape (rows, cols), (rows, cols, 1) or (rows, cols, 4)")
    return np.ascontiguousarray(a)


def _compose_on_device(mode, dst3, src3):
    """blend(dst, src) for two images of equal rows x cols on the device; returns the (rows, cols, 4) result buffer."""
    arithmetic = isinstance(mode, tuple) and len(mode) == 4
    if mode not in COMPOSE_PRE_ALPHA and not arithmetic:
        raise ValueError(f"invalid compose mode: {mode}")
    if dst3.shape[:2] != src3.shape[:2]:
        raise ValueError("canvas_compose: images must have the same rows x cols")
    rows, cols = dst3.shape[:2]
    ctx = _abi.Context.get()
    lib = ctx.lib
    bb = _bbox_arr((0, 0), (rows, cols))
    d_in, s_in = ctx.from_host(dst3), ctx.from_host(src3)
    out = ctx.alloc(max(rows * cols, 1) * 32)
    _abi._check(lib.svgr_layer_crop4(ctx.handle, out.handle, bb, d_in.handle, bb, dst3.shape[2]))  # 1 channel -> 4
    if arithmetic or mode not in (COMPOSE_OVER, COMPOSE_IN):
        k4 = np.array(mode if arithmetic else (0, 0, 0, 0), dtype=FLOAT)
        _abi._check(lib.svgr_layer_blend(ctx.handle, out.handle, bb, s_in.handle, bb, src3.shape[2], 5 if arithmetic else int(mode),
                                         k4.ctypes.data_as(_abi._P)))
    elif mode == COMPOSE_OVER:
        _abi._check(lib.svgr_layer_over(ctx.handle, out.handle, bb, s_in.handle, bb, src3.shape[2], 0))
    else:
        _abi._check(lib.svgr_layer_in(ctx.handle, out.handle, bb, s_in.handle, bb, src3.shape[2]))
    return out


def canvas_compose(mode, dst, src):
    """Compose two alpha-premultiplied images, ``src`` onto ``dst`` (canvas_compose, S:277-298): Porter-Duff OVER / OUT /
    IN / ATOP / XOR or the feComposite ``(k1, k2, k3, k4)`` arithmetic.  Images are (rows, cols[, 1 | 4]) arrays of equal
    rows x cols; a single channel is alpha and broadcasts over RGBA like numpy would."""
    d3, s3 = _image3(dst, "dst"), _image3(src, "src")
    out = _compose_on_device(mode, d3, s3)
    res = out.download(d3.shape[:2] + (4,), FLOAT)
    # numpy's broadcast decides the channels of the reference's result: OUT and IN are `src * f(dst_a)` and keep src's,
    # the other modes add a dst term and take the wider of the two.  An alpha-only result sits in every channel here.
    only_src = mode in (COMPOSE_OUT, COMPOSE_IN)
    if s3.shape[2] == 1 and (only_src or d3.shape[2] == 1):
        res = res[..., 3:]
        if np.ndim(src) == 2 and (only_src or np.ndim(dst) == 2):
            res = res[..., 0]
    return res


from functools import partial as _partial  # noqa: E402

CANVAS_COMPOSE_OVER = _partial(canvas_compose, COMPOSE_OVER)


def _blend_mode(blend):
    """The compose mode of a ``partial(canvas_compose, mode)`` (what the reference passes as ``blend``), else None."""
    if isinstance(blend, _partial) and blend.func is canvas_compose and len(blend.args) == 1 and not blend.keywords:
        return blend.args[0]
    return None


def canvas_merge_at(base, overlay, offset, blend=CANVAS_COMPOSE_OVER):
    """Blend ``overlay`` onto ``base`` at ``offset`` = (row, col), in place, clipping the touched region to [0, 1]
    (canvas_merge_at, S:304-327).  Returns ``base``, or None when nothing overlaps."""
    x, y = int(offset[0]), int(offset[1])
    b_h, b_w = base.shape[:2]
    o_h, o_w = overlay.shape[:2]
    r0, r1 = min(max(x, 0), b_h), min(max(x + o_h, 0), b_h)
    c0, c1 = min(max(y, 0), b_w), min(max(y + o_w, 0), b_w)
    if r1 <= r0 or c1 <= c0:
        return None
    region = base[r0:r1, c0:c1]
    part = overlay[r0 - x:r1 - x, c0 - y:c1 - y]
    mode = _blend_mode(blend)
    if mode is None:  # the caller's own blend function
        region[...] = np.clip(blend(region, part), 0, 1)
        return base
    out = _compose_on_device(mode, _image3(region, "base"), _image3(part, "overlay"))
    ctx = _abi.Context.get()
    n = (r1 - r0) * (c1 - c0) * 4
    _abi._check(ctx.lib.svgr_layer_clip01(ctx.handle, out.handle, n))
    res = out.download((r1 - r0, c1 - c0, 4), FLOAT)
    region[...] = res if region.ndim == 3 and region.shape[2] == 4 else res[..., 3:].reshape(region.shape)
    return base


def _as_layers(layers):
    return [Layer(_image3(image, "layer"), (int(offset[0]), int(offset[1])), True, True) for image, offset in layers]


def canvas_merge_union(layers, full=True, blend=CANVAS_COMPOSE_OVER):
    """Blend ``[(image, (row, col)), ...]`` in order into one image that holds them all; returns (image, offset)
    (canvas_merge_union, S:330-379).  ``full`` picks the reference's formulation (every layer zero-extended to the union
    first); for source-over both give the same pixels."""
    if not layers:
        raise ValueError("can not blend zero layers")
    if len(layers) == 1:
        return layers[0]
    mode = _blend_mode(blend)
    if mode is not None and mode != COMPOSE_IN:  # (Layer.compose takes IN to the intersection; here it stays on the union)
        out = Layer.compose(_as_layers(layers), mode, linear_rgb=True)
        return out.image, (out.x, out.y)
    r0 = min(int(o[0]) for _, o in layers)
    c0 = min(int(o[1]) for _, o in layers)
    r1 = max(int(o[0]) + im.shape[0] for im, o in layers)
    c1 = max(int(o[1]) + im.shape[1] for im, o in layers)
    output = None
    for image, (x, y) in layers:
        ext = np.zeros((r1 - r0, c1 - c0, 4), dtype=FLOAT)
        ext[x - r0:x - r0 + image.shape[0], y - c0:y - c0 + image.shape[1]] = image
        output = ext if output is None else blend(output, ext)
    return output, (r0, c0)


def canvas_merge_intersect(layers, blend=CANVAS_COMPOSE_OVER):
    """Blend ``[(image, (row, col)), ...]`` on the region covered by all of them; (image, offset) or None when that region
    is empty (canvas_merge_intersect, S:382-416)."""
    if not layers:
        raise ValueError("can not blend zero layers")
    if len(layers) == 1:
        return layers[0]
    r0 = max(int(o[0]) for _, o in layers)
    c0 = max(int(o[1]) for _, o in layers)
    r1 = min(int(o[0]) + im.shape[0] for im, o in layers)
    c1 = min(int(o[1]) + im.shape[1] for im, o in layers)
    if r0 >= r1 or c0 >= c1:
        return None
    crop = lambda im, o: im[r0 - int(o[0]):r1 - int(o[0]), c0 - int(o[1]):c1 - int(o[1])]  # noqa: E731
    mode = _blend_mode(blend)
    (first, f_off), *rest = layers
    output = _image3(crop(first, f_off), "layer")
    if output.shape[2] == 1:
        output = np.ascontiguousarray(np.broadcast_to(output, output.shape[:2] + (4,)))
    for image, off in rest:
        part = crop(image, off)
        output = blend(output, part) if mode is None else canvas_compose(mode, output, part)
    return output, (r0, c0)
