"""ctypes binding of libsvgr_hip.so (include/svgr.h).

There is deliberately no fallback: if the HIP library is missing or no gfx950 device is
visible, every entry point raises.  The only thing that works without a GPU is loading the
library and inspecting its symbols (used by the CPU-side ABI test).
"""
from __future__ import annotations

import ctypes as C
import os
import threading
import weakref

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "csrc", "libsvgr_hip.so")

OUT_CANVAS_F32, OUT_CANVAS_F64, OUT_MASK_F64, OUT_FILL_F64, OUT_MASKS_F64, OUT_FILLS_F64 = 0, 1, 2, 3, 4, 5
RENDER_CLIP01, RENDER_TIMED, RENDER_DETERMINISTIC, RENDER_SAME_GEOMETRY = 1, 2, 4, 8
SEG_LINE, SEG_CUBIC = 0, 1

CONVERT_PRE_TO_STRAIGHT, CONVERT_SRGB_TO_LINEAR, CONVERT_LINEAR_TO_SRGB, CONVERT_STRAIGHT_TO_PRE = 1, 2, 4, 8


class SvgrError(RuntimeError):
    """Raised for every non-zero status of the C ABI except SVGR_E_INVALID (-> ValueError)."""


class BatchDesc(C.Structure):
    _fields_ = [
        ("segs", C.c_void_p), ("seg_kind", C.c_void_p), ("n_segs", C.c_int64),
        ("path_seg_off", C.c_void_p), ("n_paths", C.c_int64),
        ("path_m6", C.c_void_p), ("path_rule", C.c_void_p), ("path_paint", C.c_void_p),
        ("viewport", C.c_int64 * 4), ("flatness", C.c_double),
    ]


class BatchStats(C.Structure):
    _fields_ = [
        ("n_edges", C.c_int64), ("path_pixels", C.c_int64), ("n_band_segs", C.c_int64),
        ("n_path_bands", C.c_int64), ("n_nonempty", C.c_int64), ("bbox_union", C.c_int64 * 4),
        ("tile_rows", C.c_int64), ("tile_cols", C.c_int64),
    ]


class Gradient(C.Structure):
    _fields_ = [
        ("kind", C.c_int), ("spread", C.c_int), ("has_gt", C.c_int), ("n_stops", C.c_int), ("excl_enabled", C.c_int),
        ("user_m6", C.c_double * 6), ("gt_m6", C.c_double * 6),
        ("p0", C.c_double * 2), ("vec", C.c_double * 2), ("vv", C.c_double),
        ("center", C.c_double * 2), ("radius", C.c_double),
        ("fcenter", C.c_double * 2), ("fradius", C.c_double), ("cd", C.c_double * 2), ("rd", C.c_double), ("a", C.c_double),
        ("frad_rd", C.c_double), ("frad2", C.c_double), ("excl_thresh", C.c_double),
        ("stop_off", C.c_void_p), ("stop_rgba", C.c_void_p),
    ]


class PatternArgs(C.Structure):  # svgr_pattern
    _fields_ = [
        ("inv_m6", C.c_double * 6), ("fwd_m6", C.c_double * 6), ("cell", C.c_double * 4),
        ("min_xy", C.c_int64 * 2), ("pat_shape", C.c_int64 * 2), ("tile_bbox", C.c_int64 * 4),
    ]


_P = C.c_void_p
_PROTOS = {
    "svgr_abi_version": (C.c_int, []),
    "svgr_hash_buffers": (C.c_int, [_P, _P, C.c_int64, C.POINTER(C.c_uint64)]),
    "svgr_tile_rows": (C.c_int, []),
    "svgr_tile_cols": (C.c_int, []),
    "svgr_last_error": (C.c_char_p, []),
    "svgr_device_count": (C.c_int, []),
    "svgr_init": (C.c_int, [C.c_int, C.POINTER(_P)]),
    "svgr_shutdown": (C.c_int, [_P]),
    "svgr_set_stream": (C.c_int, [_P, _P]),
    "svgr_sync": (C.c_int, [_P]),
    "svgr_device_name": (C.c_int, [_P, C.c_char_p, C.c_size_t]),
    "svgr_measure_begin": (C.c_int, [_P, C.c_double]),
    "svgr_measure_end": (C.c_int, [_P, C.POINTER(C.c_double)]),
    "svgr_measure_launches": (C.c_int, [C.POINTER(C.c_uint64)]),
    "svgr_buf_alloc": (C.c_int, [_P, C.c_size_t, C.POINTER(_P)]),
    "svgr_buf_wrap": (C.c_int, [_P, _P, C.c_size_t, C.POINTER(_P)]),
    "svgr_buf_free": (C.c_int, [_P, _P]),
    "svgr_buf_ptr": (_P, [_P]),
    "svgr_buf_bytes": (C.c_size_t, [_P]),
    "svgr_buf_zero": (C.c_int, [_P, _P]),
    "svgr_buf_copy": (C.c_int, [_P, _P, _P, C.c_size_t]),
    "svgr_upload": (C.c_int, [_P, _P, C.c_size_t, _P, C.c_size_t]),
    "svgr_download": (C.c_int, [_P, _P, C.c_size_t, _P, C.c_size_t]),
    "svgr_batch_create": (C.c_int, [_P, C.POINTER(BatchDesc), C.POINTER(_P)]),
    "svgr_batch_destroy": (C.c_int, [_P]),
    "svgr_batch_set_paints": (C.c_int, [_P, _P]),
    "svgr_batch_set_transforms": (C.c_int, [_P, _P]),
    "svgr_batch_set_bands": (C.c_int, [_P, C.c_int, C.c_int, C.c_int]),
    "svgr_batch_set_groups": (C.c_int, [_P, _P, C.c_int64, _P, _P]),
    "svgr_batch_set_gradients": (C.c_int, [_P, _P, C.c_int64, _P]),
    "svgr_batch_plan": (C.c_int, [_P]),
    "svgr_batch_get_stats": (C.c_int, [_P, C.POINTER(BatchStats)]),
    "svgr_batch_get_bboxes": (C.c_int, [_P, _P]),
    "svgr_batch_get_edges": (C.c_int, [_P, _P, _P, C.c_int64]),
    "svgr_batch_get_extents": (C.c_int, [_P, _P]),
    "svgr_batch_all_edges": (C.c_int, [_P, _P, _P, C.c_int64, C.POINTER(C.c_int64)]),
    "svgr_batch_render": (C.c_int, [_P, _P, C.c_int, C.c_uint]),
    "svgr_batch_draw": (C.c_int, [_P, _P, C.c_int, C.c_uint]),
    "svgr_batch_render_window": (C.c_int, [_P, _P, C.c_int, C.c_uint, _P]),
    "svgr_batch_render_windows": (C.c_int, [_P, C.c_int64, _P, C.c_int, C.c_uint, _P]),
    "svgr_batch_plan_many": (C.c_int, [_P, C.c_int64]),
    "svgr_batch_owned_rows": (C.c_int64, [_P]),
    "svgr_batch_timings": (C.c_int, [_P, C.POINTER(C.c_int), C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    "svgr_layer_over": (C.c_int, [_P, _P, _P, _P, _P, C.c_int, C.c_int]),
    "svgr_layer_crop4": (C.c_int, [_P, _P, _P, _P, _P, C.c_int]),
    "svgr_layer_in": (C.c_int, [_P, _P, _P, _P, _P, C.c_int]),
    "svgr_layer_scale": (C.c_int, [_P, _P, C.c_int64, C.c_double]),
    "svgr_layer_scale_to": (C.c_int, [_P, _P, _P, C.c_int64, C.c_double]),
    "svgr_layer_background": (C.c_int, [_P, _P, C.c_int64, _P]),
    "svgr_layer_clip01": (C.c_int, [_P, _P, C.c_int64]),
    "svgr_layer_convert": (C.c_int, [_P, _P, C.c_int64, C.c_uint]),
    "svgr_layer_convert_to": (C.c_int, [_P, _P, _P, C.c_int64, C.c_uint]),
    "svgr_layer_convert_scale_to": (C.c_int, [_P, _P, _P, C.c_int64, C.c_uint, C.c_double]),
    "svgr_layer_compose_over": (C.c_int, [_P, _P, _P, C.c_int64, _P, _P, _P, _P]),
    "svgr_layer_compose_in": (C.c_int, [_P, _P, _P, C.c_int64, _P, _P, _P, _P]),
    "svgr_layer_to_f32": (C.c_int, [_P, _P, _P, C.c_int64, C.c_int]),
    "svgr_layer_to_rgba8": (C.c_int, [_P, _P, _P, C.c_int64]),
    "svgr_layer_blend": (C.c_int, [_P, _P, _P, _P, _P, C.c_int, C.c_int, _P]),
    "svgr_layer_color_matrix": (C.c_int, [_P, _P, C.c_int64, _P]),
    "svgr_layer_morphology": (C.c_int, [_P, _P, _P, C.c_int64, C.c_int64, C.c_int64, C.c_int64, C.c_int]),
    "svgr_layer_luminance": (C.c_int, [_P, _P, _P, C.c_int64]),
    "svgr_gradient_fill": (C.c_int, [_P, C.POINTER(Gradient), _P, _P, _P]),
    "svgr_gradient_eval": (C.c_int, [_P, C.POINTER(Gradient), _P, C.c_int64, _P]),
    "svgr_pattern_fill": (C.c_int, [_P, C.POINTER(PatternArgs), _P, _P, _P, _P]),
    "svgr_layer_convolve": (C.c_int, [_P, _P, _P, C.c_int64, C.c_int64, _P, C.c_int64, C.c_int64]),
    "svgr_layer_convolve_ops": (C.c_int, [_P, _P, _P, C.c_int64, C.c_int64, _P, C.c_int64, C.c_int64, C.c_uint]),
    "svgr_path_stroke": (C.c_int, [_P, _P, _P, C.c_int64, C.c_double, C.c_int, C.c_int, C.POINTER(_P)]),
    "svgr_stroke_out_counts": (C.c_int, [_P, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    "svgr_stroke_out_copy": (C.c_int, [_P, _P, _P, _P]),
    "svgr_stroke_out_free": (None, [_P]),
}
EXPORTS = tuple(_PROTOS)

_lib = None
_lock = threading.Lock()


def load_library():
    """dlopen libsvgr_hip.so and declare prototypes; raises if it has not been built."""
    global _lib
    with _lock:
        if _lib is not None:
            return _lib
        if not os.path.exists(LIB_PATH):
            raise SvgrError(
                f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(there is no CPU fallback)"
            )
        lib = C.CDLL(LIB_PATH)
        for name, (res, args) in _PROTOS.items():
            fn = getattr(lib, name)
            fn.restype = res
            fn.argtypes = args
        _lib = lib
        return lib


def tile_rows() -> int:
    """Band height of the built library (row granularity of Batch.set_bands)."""
    return int(load_library().svgr_tile_rows())


def tile_cols() -> int:
    """Tile width of the built library."""
    return int(load_library().svgr_tile_cols())


def _check(rc: int):
    if rc == 0:
        return
    msg = load_library().svgr_last_error().decode("utf-8", "replace")
    if rc == -1:
        raise ValueError(msg)
    raise SvgrError(f"svgr status {rc}: {msg}")


class Context:
    """One per device (svgr_ctx)."""

    _by_device: dict[int, "Context"] = {}

    def __init__(self, device: int = 0):
        self.lib = load_library()
        h = _P()
        _check(self.lib.svgr_init(device, C.byref(h)))
        self.handle = h
        self.device = device
        self._fin = weakref.finalize(self, self.lib.svgr_shutdown, h)

    _default: "Context | None" = None   # the context of $SVGR_DEVICE / $LOCAL_RANK, resolved once (a document render asks 200 times)

    @classmethod
    def get(cls, device: int | None = None) -> "Context":
        if device is None:
            ctx = cls._default
            if ctx is not None:
                return ctx
            device = int(os.environ.get("SVGR_DEVICE") or os.environ.get("LOCAL_RANK") or "0")
            ctx = cls._default = cls.get(device)
            return ctx
        ctx = cls._by_device.get(device)
        if ctx is None:
            ctx = cls._by_device[device] = Context(device)
        return ctx

    def name(self) -> str:
        buf = C.create_string_buffer(160)
        _check(self.lib.svgr_device_name(self.handle, buf, 160))
        return buf.value.decode()

    def sync(self):
        _check(self.lib.svgr_sync(self.handle))

    # -- measurement helpers (bench.py) -------------------------------------------------------
    def measure_begin(self, hold_ms: float = 0.0):
        _check(self.lib.svgr_measure_begin(self.handle, float(hold_ms)))

    def measure_end(self) -> float:
        ms = C.c_double()
        _check(self.lib.svgr_measure_end(self.handle, C.byref(ms)))
        return ms.value

    def launches(self) -> int:
        n = C.c_uint64()
        _check(self.lib.svgr_measure_launches(C.byref(n)))
        return int(n.value)

    def set_stream(self, hip_stream: int):
        _check(self.lib.svgr_set_stream(self.handle, _P(hip_stream)))

    # -- buffers -------------------------------------------------------------------------
    def alloc(self, nbytes: int) -> "DeviceBuffer":
        h = _P()
        _check(self.lib.svgr_buf_alloc(self.handle, nbytes, C.byref(h)))
        return DeviceBuffer(self, h, nbytes)

    def wrap(self, device_ptr: int, nbytes: int) -> "DeviceBuffer":
        h = _P()
        _check(self.lib.svgr_buf_wrap(self.handle, _P(device_ptr), nbytes, C.byref(h)))
        return DeviceBuffer(self, h, nbytes)

    def from_host(self, arr: np.ndarray) -> "DeviceBuffer":
        arr = np.ascontiguousarray(arr)
        buf = self.alloc(arr.nbytes)
        buf.upload(arr)
        return buf


class DeviceBuffer:
    # (freed by `__del__`, not by a weakref.finalize: a document's render makes and drops 150 of these, and a finalizer object
    #  apiece was 0.3 ms of it.  svgr_buf_free does not touch the context -- the block goes back to the library's pool, which
    #  outlives every context --, so the order of the interpreter's teardown does not matter)
    __slots__ = ("ctx", "handle", "nbytes", "_free", "_parent", "__weakref__")   # (`_parent`: the buffer a view made by `wrap` lives in)

    def __init__(self, ctx: Context, handle, nbytes: int):
        self.ctx, self.handle, self.nbytes = ctx, handle, nbytes
        self._free = ctx.lib.svgr_buf_free

    @property
    def ptr(self) -> int:
        return int(self.ctx.lib.svgr_buf_ptr(self.handle) or 0)

    def zero(self):
        _check(self.ctx.lib.svgr_buf_zero(self.ctx.handle, self.handle))

    def upload(self, arr: np.ndarray, offset: int = 0):
        arr = np.ascontiguousarray(arr)
        _check(self.ctx.lib.svgr_upload(self.ctx.handle, self.handle, offset, arr.ctypes.data_as(_P), arr.nbytes))

    def download(self, shape, dtype, offset: int = 0) -> np.ndarray:
        out = np.empty(shape, dtype=dtype)
        _check(self.ctx.lib.svgr_download(self.ctx.handle, self.handle, offset, out.ctypes.data_as(_P), out.nbytes))
        return out

    def free(self):
        h = self.handle
        if h is not None:
            self.handle = None
            try:
                self._free(None, h)
            except Exception:  # noqa: BLE001  (interpreter teardown)
                pass

    __del__ = free


def _i64x4(v):
    return (C.c_int64 * 4)(*[int(x) for x in v])


def ptr(arr) -> int:
    """Address of a numpy array's data (``arr.ctypes`` builds a helper object on every access: 4 us that add up over a
    scene's hundreds of small arrays)."""
    return arr.__array_interface__["data"][0]


class Batch:
    """svgr_batch: a paint-ordered list of paths resident in HBM."""

    def __init__(self, ctx: Context, segs, seg_kind, path_seg_off, path_m6, path_rule, path_paint,
                 viewport=None, flatness: float = 0.1):
        self.ctx = ctx
        lib = ctx.lib
        segs = np.ascontiguousarray(segs, dtype=np.float64).reshape(-1, 8)
        seg_kind = np.ascontiguousarray(seg_kind, dtype=np.uint8).reshape(-1)
        path_seg_off = np.ascontiguousarray(path_seg_off, dtype=np.int64).reshape(-1)
        n_paths = len(path_seg_off) - 1
        path_m6 = np.ascontiguousarray(path_m6, dtype=np.float64).reshape(-1, 6)
        path_rule = np.ascontiguousarray(path_rule, dtype=np.uint8).reshape(-1)
        path_paint = np.ascontiguousarray(path_paint, dtype=np.float64).reshape(-1, 4)
        if len(seg_kind) != len(segs) or len(path_m6) != n_paths or len(path_rule) != n_paths or len(path_paint) != n_paths:
            raise ValueError("inconsistent batch arrays")
        d = BatchDesc()
        d.segs = ptr(segs)
        d.seg_kind = ptr(seg_kind)
        d.n_segs = len(segs)
        d.path_seg_off = ptr(path_seg_off)
        d.n_paths = n_paths
        d.path_m6 = ptr(path_m6)
        d.path_rule = ptr(path_rule)
        d.path_paint = ptr(path_paint)
        d.viewport = _i64x4(viewport if viewport is not None else (0, 0, 0, 0))
        d.flatness = flatness
        h = _P()
        _check(lib.svgr_batch_create(ctx.handle, C.byref(d), C.byref(h)))
        self.handle = h
        self.n_paths = n_paths
        self.n_segs = len(segs)
        self._fin = weakref.finalize(self, lib.svgr_batch_destroy, h)
        self._stats = None

    def destroy(self):
        """Free the batch now.  (Dropping the last reference does the same: callers that hand out lazy views of a batch -- the
        hulls of Path.mask / Scene.render -- simply let go of it.)  A destroyed batch refuses every later call."""
        self._fin()
        self.handle = None

    def plan(self) -> "BatchStats":
        _check(self.ctx.lib.svgr_batch_plan(self.handle))
        st = BatchStats()
        _check(self.ctx.lib.svgr_batch_get_stats(self.handle, C.byref(st)))
        self._stats = st
        return st

    @staticmethod
    def plan_many(batches) -> None:
        """svgr_batch_plan for all of `batches` behind one wait (svgr_batch_plan_many); their `stats` are set."""
        batches = list(batches)
        if not batches:
            return
        arr = (_P * len(batches))(*[b.handle for b in batches])
        _check(batches[0].ctx.lib.svgr_batch_plan_many(arr, len(batches)))
        for b in batches:
            st = BatchStats()
            _check(b.ctx.lib.svgr_batch_get_stats(b.handle, C.byref(st)))
            b._stats = st

    @property
    def stats(self) -> BatchStats:
        if self._stats is None:
            st = BatchStats()
            if self.ctx.lib.svgr_batch_get_stats(self.handle, C.byref(st)) == 0:   # (planned already, e.g. by draw())
                self._stats = st
            else:
                self.plan()
        return self._stats

    def bboxes(self) -> np.ndarray:
        out = np.empty((self.n_paths, 4), dtype=np.int32)
        _check(self.ctx.lib.svgr_batch_get_bboxes(self.handle, out.ctypes.data_as(_P)))
        return out

    def edges(self):
        n = int(self.stats.n_edges)
        edges = np.empty((n, 2, 2), dtype=np.float64)
        edge_path = np.empty(n, dtype=np.int32)
        _check(self.ctx.lib.svgr_batch_get_edges(self.handle, edges.ctypes.data_as(_P), edge_path.ctypes.data_as(_P), n))
        return edges, edge_path

    def all_edges(self):
        """Every flattened edge, also those off the viewport (what the reference builds Path.mask's hull from)."""
        n = C.c_int64()
        _check(self.ctx.lib.svgr_batch_all_edges(self.handle, None, None, 0, C.byref(n)))
        edges = np.empty((n.value, 2, 2), dtype=np.float64)
        edge_path = np.empty(n.value, dtype=np.int32)
        if n.value:
            _check(self.ctx.lib.svgr_batch_all_edges(self.handle, edges.ctypes.data_as(_P), edge_path.ctypes.data_as(_P), n.value, C.byref(n)))
        return edges, edge_path

    def set_bands(self, rank: int, world: int, strip_bands: int = 1):
        """Shard by interleaved strips of `strip_bands` bands; call plan() again afterwards."""
        _check(self.ctx.lib.svgr_batch_set_bands(self.handle, rank, world, strip_bands))
        self._stats = None

    def set_groups(self, path_group, group_clip_src, group_opacity):
        """Isolated groups (CLIP / OPACITY over a GROUP of solid fills): see svgr_batch_set_groups; call before plan()."""
        pg = np.ascontiguousarray(path_group, dtype=np.int32).reshape(self.n_paths)
        cs = np.ascontiguousarray(group_clip_src, dtype=np.int32).reshape(-1)
        op = np.ascontiguousarray(group_opacity, dtype=np.float64).reshape(-1)
        if len(cs) != len(op):
            raise ValueError("one clip source and one opacity per group")
        _check(self.ctx.lib.svgr_batch_set_groups(self.handle, C.c_void_p(ptr(pg)), len(cs), C.c_void_p(ptr(cs)), C.c_void_p(ptr(op))))
        self._stats = None

    def extents(self) -> np.ndarray:
        """(n_paths, 4) doubles {min row, min col, max row, max col} of every path's flattened points, unclipped
        (svgr_batch_get_extents: between plan() and the first render)."""
        out = np.empty((self.n_paths, 4), dtype=np.float64)
        _check(self.ctx.lib.svgr_batch_get_extents(self.handle, C.c_void_p(ptr(out))))
        return out

    def set_gradients(self, path_grad, grads):
        """Gradient paints inside the batch: path_grad[p] = index into `grads` (a list of `Gradient` structs, one per
        gradient-filled path) or -1; see svgr_batch_set_gradients.  Call before plan()."""
        pg = np.ascontiguousarray(path_grad, dtype=np.int32).reshape(self.n_paths)
        arr = (Gradient * max(len(grads), 1))(*grads)
        _check(self.ctx.lib.svgr_batch_set_gradients(self.handle, C.c_void_p(ptr(pg)), len(grads), C.cast(arr, _P)))
        self._stats = None

    def set_paints(self, paints):
        paints = np.ascontiguousarray(paints, dtype=np.float64).reshape(self.n_paths, 4)
        _check(self.ctx.lib.svgr_batch_set_paints(self.handle, paints.ctypes.data_as(_P)))

    def set_transforms(self, path_m6):
        """New transforms for the same geometry (svgr_batch_set_transforms): the plan is void until plan() runs again."""
        m6 = np.ascontiguousarray(path_m6, dtype=np.float64).reshape(self.n_paths, 6)
        _check(self.ctx.lib.svgr_batch_set_transforms(self.handle, m6.ctypes.data_as(_P)))
        self._stats = None

    def owned_rows(self) -> int:
        return int(self.ctx.lib.svgr_batch_owned_rows(self.handle))

    def render(self, out: DeviceBuffer, kind: int, flags: int = 0, window=None):
        """`window` (row0, col0, rows, cols): only that part of the canvas, into a buffer of its size (svgr_batch_render_window)."""
        if window is None:
            _check(self.ctx.lib.svgr_batch_render(self.handle, out.handle, kind, flags))
        else:
            w = (C.c_int32 * 4)(*[int(v) for v in window])
            _check(self.ctx.lib.svgr_batch_render_window(self.handle, out.handle, kind, flags, w))

    def draw(self, out: DeviceBuffer, kind: int, flags: int = 0):
        """svgr_batch_draw: plan (if the batch has no valid plan) and render behind ONE wait -- a frame with new geometry, the
        reference's only mode (S:948-957).  The picture is in `out` and the stream has drained when the call returns."""
        _check(self.ctx.lib.svgr_batch_draw(self.handle, out.handle, kind, flags))
        self._stats = None   # (whatever plan() returned before describes other geometry: `stats` asks again)

    def render_windows(self, outs, kind: int, windows, flags: int = 0):
        """svgr_batch_render_windows: `windows[i]` (row0, col0, rows, cols) into `outs[i]`, all from one geometry pass, side by side."""
        n = len(outs)
        if n == 0:
            return
        arr = (_P * n)(*[o.handle for o in outs])
        w = (C.c_int32 * (4 * n))(*[int(v) for win in windows for v in win])
        _check(self.ctx.lib.svgr_batch_render_windows(self.handle, n, arr, kind, flags, w))

    def render_masks(self):
        """SVGR_OUT_MASKS_F64: Path.mask of every path of the batch in one launch.  Returns (buffer, offsets, bboxes):
        path p's (rows_p, cols_p) doubles start `offsets[p]` doubles into `buffer`."""
        bb = self.bboxes()
        area = np.where((bb[:, 2] > 0) & (bb[:, 3] > 0), bb[:, 2].astype(np.int64) * bb[:, 3], 0)
        offs = np.concatenate([[0], np.cumsum(area)]).astype(np.int64)
        buf = self.ctx.alloc(max(int(offs[-1]) * 8, 8))
        self.render(buf, OUT_MASKS_F64)
        return buf, offs, bb

    def render_fills(self):
        """SVGR_OUT_FILLS_F64: Path.fill of every path of the batch (its own solid paint) in one launch.  Returns (buffer,
        offsets, bboxes): path p's (rows_p, cols_p, 4) doubles start `4 * offsets[p]` doubles into `buffer`."""
        bb = self.bboxes()
        area = np.where((bb[:, 2] > 0) & (bb[:, 3] > 0), bb[:, 2].astype(np.int64) * bb[:, 3], 0)
        offs = np.concatenate([[0], np.cumsum(area)]).astype(np.int64)
        buf = self.ctx.alloc(max(int(offs[-1]) * 32, 32))
        self.render(buf, OUT_FILLS_F64)
        return buf, offs, bb

    def timings(self):
        n = C.c_int()
        tot, geo, tile = C.c_double(), C.c_double(), C.c_double()
        _check(self.ctx.lib.svgr_batch_timings(self.handle, C.byref(n), C.byref(tot), C.byref(geo), C.byref(tile)))
        return dict(n=n.value, ms_total=tot.value, ms_geometry=geo.value, ms_tile=tile.value)


def hash_buffers(ptrs: np.ndarray, sizes: np.ndarray) -> int:
    """svgr_hash_buffers (host only): 64-bit hash over the bytes of the buffers (ptrs uint64, sizes int64)."""
    lib = load_library()
    out = C.c_uint64()
    n = len(ptrs)
    _check(lib.svgr_hash_buffers(ptrs.ctypes.data_as(_P), sizes.ctypes.data_as(_P), n, C.byref(out)))
    return int(out.value)


def path_stroke(seg_types, seg_params, subpath_sizes, width: float, linecap: int, linejoin: int):
    """svgr_path_stroke (host C++, no GPU involved): (types, params (n, 8), sizes) of the stroke outline."""
    lib = load_library()
    seg_types = np.ascontiguousarray(seg_types, dtype=np.int32)
    seg_params = np.ascontiguousarray(seg_params, dtype=np.float64).reshape(-1, 8)
    subpath_sizes = np.ascontiguousarray(subpath_sizes, dtype=np.int32)
    if int(subpath_sizes.sum()) != len(seg_types) or len(seg_params) != len(seg_types):
        raise ValueError("segment arrays do not match the subpath sizes")
    out = _P()
    rc = lib.svgr_path_stroke(seg_types.ctypes.data_as(_P), seg_params.ctypes.data_as(_P), subpath_sizes.ctypes.data_as(_P),
                              len(subpath_sizes), float(width), int(linecap), int(linejoin), C.byref(out))
    if rc != 0:
        why = {-1: "unsupported segment type or bad cap / join", -3: "out of memory"}.get(
            rc, "the offset of a cubic does not converge (degenerate control points)")
        raise ValueError(f"svgr_path_stroke failed ({rc}): {why}")
    try:
        n, ns = C.c_int64(), C.c_int64()
        lib.svgr_stroke_out_counts(out, C.byref(n), C.byref(ns))
        types = np.empty(n.value, dtype=np.int32)
        params = np.empty((n.value, 8), dtype=np.float64)
        sizes = np.empty(ns.value, dtype=np.int32)
        lib.svgr_stroke_out_copy(out, types.ctypes.data_as(_P), params.ctypes.data_as(_P), sizes.ctypes.data_as(_P))
    finally:
        lib.svgr_stroke_out_free(out)
    return types, params, sizes
