"""Host-side geometry types with the reference's names and call signatures.

Mirrors (reference = svgrasterize.py):
    Transform    S:509-570     affine 3x3 wrapper (points are row vectors, ``points @ M.T + b``)
    ConvexHull   S:1963-2029   returned by every Path.mask / Path.fill; built lazily here
    Path         S:896-1103    ``subpaths`` in the reference's own segment format; ``mask`` and
                               ``fill`` run on the GPU through the C ABI (no CPU fallback)
Only what the hot path needs is here: no SVG path-data parser beyond a small convenience one,
no stroker.
"""
from __future__ import annotations

import math
import warnings
from typing import Callable, Iterable, Sequence

import numpy as np

from . import _abi
from ._state import STATE
from .layer import Layer

FLOAT = np.float64

PATH_LINE, PATH_QUAD, PATH_CUBIC, PATH_ARC, PATH_CLOSED, PATH_UNCLOSED = 0, 1, 2, 3, 4, 5
PATH_LINES = {PATH_LINE, PATH_CLOSED, PATH_UNCLOSED}
PATH_FILL_NONZERO = "nonzero"
PATH_FILL_EVENODD = "evenodd"
STROKE_JOIN_MITER, STROKE_JOIN_ROUND, STROKE_JOIN_BEVEL = "miter", "round", "bevel"  # S:876-878
STROKE_CAP_BUTT, STROKE_CAP_ROUND, STROKE_CAP_SQUARE = "butt", "round", "square"      # S:879-881
_RULES = {None: 0, PATH_FILL_NONZERO: 0, PATH_FILL_EVENODD: 1}

FLATNESS = 0.1  # S:955


# --------------------------------------------------------------------------------------
# Transform
# --------------------------------------------------------------------------------------
_INV_MEMO: dict = {}


class Transform:
    __slots__ = ["m", "_m_inv", "_m6", "_key"]

    def __init__(self, matrix=None, matrix_inv=None):
        if matrix is None:
            self.m = np.identity(3)
            self._m_inv = self.m
        else:
            self.m = matrix
            self._m_inv = matrix_inv
        self._m6 = self._key = None   # (made on first use: the leaves of a group share their accumulated transform)

    def __matmul__(self, other: "Transform") -> "Transform":
        # (np.dot: the product the reference's `self.m @ other.m` (S:516-518) is -- the same BLAS call, checked bit for bit on
        #  200 000 random pairs -- without the matmul ufunc's set-up, which was two thirds of the 1.6 us; a document's walk makes
        #  a thousand of these)
        return Transform(np.dot(self.m, other.m))

    @property
    def invert(self) -> "Transform":
        if self._m_inv is None:
            # (np.linalg.inv like the reference, S:520-523; the leaves of a group share their accumulated matrix, so the
            #  20 us LAPACK call is looked up by the matrix's bytes: same input, same bits)
            key = self.m.tobytes()
            inv = _INV_MEMO.get(key)
            if inv is None:
                if len(_INV_MEMO) > 4096:
                    _INV_MEMO.clear()
                inv = _INV_MEMO[key] = np.linalg.inv(self.m)
            self._m_inv = inv
        return Transform(self._m_inv, self.m)

    def __call__(self, points):
        """Host evaluation (used for hull / bbox maths, not for rasterisation: the device
        applies the matrix itself with the same fma form, see csrc/svgr_core.h)."""
        points = np.asarray(points, dtype=FLOAT)
        if len(points) == 0:
            return points
        return points @ self.m[:2, :2].T + self.m[:2, 2]

    def apply(self) -> Callable:
        """The map as a plain function of point arrays, matrix slices taken once (S:536-539)."""
        lin, shift = self.m[:2, :2].T, self.m[:2, 2]
        return lambda points: points @ lin + shift

    def _chain(self, rhs) -> "Transform":
        return Transform(self.m @ np.array(rhs, dtype=FLOAT))

    def matrix(self, m00, m01, m02, m10, m11, m12) -> "Transform":
        return self._chain([[m00, m01, m02], [m10, m11, m12], [0, 0, 1]])

    def translate(self, tx, ty) -> "Transform":
        return self._chain([[1, 0, tx], [0, 1, ty], [0, 0, 1]])

    def scale(self, sx, sy=None) -> "Transform":
        sy = sx if sy is None else sy
        return self._chain([[sx, 0, 0], [0, sy, 0], [0, 0, 1]])

    def rotate(self, angle) -> "Transform":
        c, s = math.cos(angle), math.sin(angle)
        return self._chain([[c, -s, 0], [s, c, 0], [0, 0, 1]])

    def skew(self, ax, ay) -> "Transform":
        return self._chain([[1, math.tan(ax), 0], [math.tan(ay), 1, 0], [0, 0, 1]])

    def no_translate(self) -> "Transform":
        m = self.m.copy()
        m[0, 2] = 0
        m[1, 2] = 0
        return Transform(m)

    def m6(self) -> np.ndarray:
        """The six numbers the C ABI takes: rows 0-1 of the matrix."""
        if self._m6 is None:
            m = self.m
            # an OWNED copy, taken together with the key: a caller who edits `m` in place afterwards cannot make the six numbers
            # the batches see and the bytes the memos are keyed by disagree (a Transform is a value; its methods return new ones)
            m6 = np.array(np.asarray(m)[:2, :], dtype=FLOAT).reshape(6)
            m6.flags.writeable = False   # (handed out to many leaves)
            self._key = m6.tobytes()
            self._m6 = m6
        return self._m6

    def key(self) -> bytes:
        """The matrix as bytes (memo keys): the bytes of `m6()`, made in the same step."""
        if self._key is None:
            self.m6()
        return self._key

    def __repr__(self) -> str:
        return str(np.around(self.m, 4).tolist()[:2])


# --------------------------------------------------------------------------------------
# ConvexHull (lazy)
# --------------------------------------------------------------------------------------
def _graham(points: list) -> list:
    """Monotone-chain convex hull over (row, col) tuples, same tie rules as S:1977-1992."""

    def turn(p, q, r):
        return (q[0] - p[0]) * (r[1] - p[1]) - (r[0] - p[0]) * (q[1] - p[1])

    def half(seq):
        out: list = []
        for p in seq:
            while len(out) > 1 and turn(out[-2], out[-1], p) <= 0:
                out.pop()
            if not out or out[-1] != p:
                out.append(p)
        return out

    pts = sorted(points)
    left = half(pts)
    right = half(reversed(pts))
    left.extend(right[1:-1])
    return left


class ConvexHull:
    """Convex hull of flattened path points in presentation space.

    The reference builds it eagerly from every flattened endpoint on each mask()/fill()
    (S:993); almost no caller reads it, so here the point source is a thunk that downloads the
    device edge list only when ``points`` is first touched."""

    __slots__ = ["_points", "_source"]

    def __init__(self, points=None, _source: Callable[[], np.ndarray] | None = None):
        self._points = None
        self._source = _source
        if points is not None:
            if isinstance(points, np.ndarray):
                points = points.reshape(-1, 2).tolist()
            self._points = _graham([list(p) for p in points])

    @property
    def points(self) -> list:
        if self._points is None:
            src = self._source() if self._source is not None else np.zeros((0, 2))
            self._points = _graham(np.asarray(src, dtype=FLOAT).reshape(-1, 2).tolist())
            self._source = None
        return self._points

    @classmethod
    def merge(cls, hulls: Iterable["ConvexHull"]) -> "ConvexHull":
        hulls = list(hulls)

        def gather():
            pts = []
            for h in hulls:
                pts.extend(h.points)
            return np.array(pts, dtype=FLOAT).reshape(-1, 2)

        return cls(_source=gather)

    def bbox(self, transform: Transform):
        points = transform.invert(np.array(self.points))
        min_x, min_y = points.min(axis=0)
        max_x, max_y = points.max(axis=0)
        return [min_x, min_y, max_x - min_x, max_y - min_y]

    def bbox_transform(self, transform: Transform) -> Transform:
        x, y, w, h = self.bbox(transform)
        if w <= 0 and h <= 0:
            return transform
        return transform.translate(x, y).scale(w, h)

    def path(self) -> "Path":
        """The hull outline as a closed polygon path (S:2025-2029)."""
        pts = self.points
        edges = [(PATH_LINE, pair) for pair in zip(pts, pts[1:])] + [(PATH_CLOSED, [pts[-1], pts[0]])]
        return Path([edges])


# --------------------------------------------------------------------------------------
# curve conversions that stay on the host
# --------------------------------------------------------------------------------------
_QUAD_TO_CUBIC = np.array([[1, 0, 0], [1.0 / 3, 2.0 / 3, 0], [0, 2.0 / 3.0, 1.0 / 3], [0, 0, 1]], dtype=FLOAT)


def quad_to_cubic(points) -> np.ndarray:
    """Degree elevation of a quadratic Bezier (S:2052-2054, S:2182-2184)."""
    return _QUAD_TO_CUBIC @ np.asarray(points, dtype=FLOAT)


def arc_to_cubics(center, rx, ry, phi, eta, eta_delta) -> np.ndarray:
    """Elliptical arc (centre parametrisation) -> cubic Beziers of at most pi/4 each (S:2355-2394)."""
    rot = np.array([[math.cos(phi), -math.sin(phi)], [math.sin(phi), math.cos(phi)]])

    def point(a):
        return rot @ [rx * math.cos(a), ry * math.sin(a)] + center

    def deriv(a):
        return rot @ [-rx * math.sin(a), ry * math.cos(a)]

    count = math.ceil(abs(eta_delta) / (math.pi / 4))
    etas = np.linspace(eta, eta + eta_delta, count + 1)
    out = []
    for e1, e2 in zip(etas, etas[1:]):
        root = math.sqrt(4 + 3 * math.tan((e2 - e1) / 2) ** 2)
        alpha = math.sin(e2 - e1) * (root - 1) / 3
        p0, p3 = point(e1), point(e2)
        out.append([p0, p0 + alpha * deriv(e1), p3 - alpha * deriv(e2), p3])
    return np.array(out)


# --------------------------------------------------------------------------------------
# Path
# --------------------------------------------------------------------------------------
class Path:
    """Rendering unit; ``subpaths`` uses the reference's segment tuples (S:899-907)."""

    __slots__ = ["subpaths", "_packed", "_user_box"]

    def __init__(self, subpaths):
        self.subpaths = subpaths
        self._packed = None
        self._user_box = False

    def __iter__(self):
        return iter(self.subpaths)

    def __bool__(self) -> bool:
        return bool(self.subpaths)

    def is_empty(self) -> bool:
        return not bool(self.subpaths)

    def transform(self, transform: Transform) -> "Path":
        """The path with ``transform`` applied on the host (S:1182-1202); arcs become cubics first.  Rendering does not
        use this: ``mask`` / ``fill`` take the transform and apply it on the device."""
        out = []
        for sub in self.subpaths:
            if not sub:
                continue
            moved = []
            for kind, params in sub:
                if kind == PATH_ARC:
                    moved.extend((PATH_CUBIC, c.tolist()) for c in transform(arc_to_cubics(*params)))
                else:
                    moved.append((kind, transform(np.array(params)).tolist()))
            out.append(moved)
        return Path(out)

    def to_svg(self) -> str:
        """Path data, one line per subpath, numbers in ``%g`` (S:1204-1251): a segment names its command only when the
        kind of segment changes, the first one is preceded by a moveto to its start."""
        lines = []
        for sub in self.subpaths:
            if not sub:
                continue
            words, last = [], None
            for kind, params in sub:
                if kind == PATH_CLOSED:
                    words.append("Z ")
                    last = None
                    continue
                if kind == PATH_UNCLOSED:
                    last = None
                    continue
                if kind not in (PATH_LINE, PATH_QUAD, PATH_CUBIC, PATH_ARC):
                    raise ValueError("unhandled path type: `{cmd}`")
                pieces = arc_to_cubics(*params) if kind == PATH_ARC else [params]
                for pts in pieces:
                    (x0, y0), rest = pts[0], pts[1:]
                    if last != kind:
                        if last is None:
                            words.append(f"M{x0:g},{y0:g} ")
                        if kind == PATH_LINE:
                            words.append("L" if last is not None else "")
                        else:
                            words.append("Q" if kind == PATH_QUAD else "C")
                    words.append("".join(f"{x:g},{y:g} " for x, y in rest))
                    last = PATH_CUBIC if kind == PATH_ARC else kind
            lines.append("".join(words))
        return "\n".join(lines)

    def __repr__(self) -> str:
        if not self.subpaths:
            return "EMPTY"
        pts = lambda coords: " ".join(f"{x:.4g},{y:.4g}" for x, y in coords)  # noqa: E731
        names = {PATH_LINE: "LINE", PATH_CUBIC: "CUBIC", PATH_QUAD: "QUAD"}
        rows = []
        for sub in self.subpaths:
            for kind, params in sub:
                if kind in names:
                    rows.append(f"{names[kind]} {pts(params)}")
                elif kind == PATH_ARC:
                    center, rx, ry, phi, eta, eta_delta = params
                    rows.append(f"ARC {pts([center])} {rx:.4g} {ry:.4g} {phi:.3g} {eta:.3g} {eta_delta:.3g}")
                elif kind == PATH_CLOSED:
                    rows.append("CLOSE")
        return "\n".join(rows)

    # -- construction helpers ------------------------------------------------------------
    @classmethod
    def from_arrays(cls, lines, cubics) -> "Path":
        """Build a path from (N,2,2) lines and (M,4,2) cubics (scene dumps, synthetic scenes)."""
        lines = np.asarray(lines, dtype=FLOAT).reshape(-1, 2, 2)
        cubics = np.asarray(cubics, dtype=FLOAT).reshape(-1, 4, 2)
        path = cls([[(PATH_LINE, l) for l in lines] + [(PATH_CUBIC, c) for c in cubics]])
        segs = np.zeros((len(lines) + len(cubics), 8))
        segs[: len(lines), :4] = lines.reshape(-1, 4)
        segs[len(lines):] = cubics.reshape(-1, 8)
        kinds = np.zeros(len(segs), dtype=np.uint8)
        kinds[len(lines):] = _abi.SEG_CUBIC
        path._packed = (segs, kinds)
        return path

    @classmethod
    def from_segments(cls, seg_types, seg_params, subpath_sizes) -> "Path":
        """Rebuild ``subpaths`` from flat arrays (type code, 8 numbers per segment)."""
        subpaths, k = [], 0
        for n in subpath_sizes:
            sub = []
            for _ in range(int(n)):
                t, p = int(seg_types[k]), np.asarray(seg_params[k], dtype=FLOAT)
                if t in PATH_LINES:
                    sub.append((t, p[:4].reshape(2, 2)))
                elif t == PATH_QUAD:
                    sub.append((t, p[:6].reshape(3, 2)))
                elif t == PATH_CUBIC:
                    sub.append((t, p[:8].reshape(4, 2)))
                elif t == PATH_ARC:
                    sub.append((t, (p[:2].copy(), p[2], p[3], p[4], p[5], p[6])))
                else:
                    raise ValueError(f"unsupported path type: `{t}`")
                k += 1
            subpaths.append(sub)
        return cls(subpaths)

    def packed(self):
        """(segs (n, 8), kinds (n,)) in user space: explicit lines first, then cubics, as the
        reference gathers ``lines_defs`` / ``cubics_defs`` (S:930-945)."""
        if self._packed is None:
            lines, cubics = [], []
            for sub in self.subpaths:
                for seg in sub:
                    t = seg[0]
                    if t in PATH_LINES:
                        lines.append(np.asarray(seg[1], dtype=FLOAT).reshape(4))
                    elif t == PATH_CUBIC:
                        cubics.append(np.asarray(seg[1], dtype=FLOAT).reshape(8))
                    elif t == PATH_QUAD:
                        cubics.append(quad_to_cubic(seg[1]).reshape(8))
                    elif t == PATH_ARC:
                        cubics.extend(c.reshape(8) for c in arc_to_cubics(*seg[1]))
                    else:
                        raise ValueError(f"unsupported path type: `{t}`")
            segs = np.zeros((len(lines) + len(cubics), 8))
            if lines:
                segs[: len(lines), :4] = np.array(lines)
            if cubics:
                segs[len(lines):] = np.array(cubics)
            kinds = np.zeros(len(segs), dtype=np.uint8)
            kinds[len(lines):] = _abi.SEG_CUBIC
            self._packed = (segs, kinds)
        return self._packed

    def user_box(self):
        """(x0, y0, x1, y1) of the control points in user space (every curve lies inside its control polygon); None: no
        segment, or a coordinate that is not finite.  Computed once."""
        if self._user_box is False:
            segs, kinds = self.packed()
            box = None
            if len(segs):
                pts = segs.reshape(-1, 4, 2)
                cubic = kinds != 0
                xs = np.concatenate([pts[:, :2, 0].ravel(), pts[cubic][:, 2:, 0].ravel()])
                ys = np.concatenate([pts[:, :2, 1].ravel(), pts[cubic][:, 2:, 1].ravel()])
                box = (float(xs.min()), float(ys.min()), float(xs.max()), float(ys.max()))
                if not all(math.isfinite(v) for v in box):
                    box = None
            self._user_box = box
        return self._user_box

    # -- the hot path ---------------------------------------------------------------------
    def _single_batch(self, transform: Transform, fill_rule, viewport, paint=None):
        if fill_rule not in _RULES:
            raise ValueError(f"Invalid fill rule: {fill_rule}")
        segs, kinds = self.packed()
        if len(segs) == 0:
            return None
        ctx = _abi.Context.get()
        vp = None
        if viewport is not None:
            vp = [int(v) for v in viewport]
        batch = _abi.Batch(
            ctx, segs, kinds, [0, len(segs)], transform.m6(), [_RULES[fill_rule]],
            [paint if paint is not None else np.zeros(4)], viewport=vp, flatness=FLATNESS,
        )
        batch.plan()
        bb = batch.bboxes()[0]
        if bb[2] <= 0 or bb[3] <= 0:
            batch.destroy()
            return None
        return ctx, batch, bb

    def mask(self, transform: Transform, fill_rule: str | None = None, viewport=None):
        """Render path as a mask (alpha channel only image), S:922-993.

        Returns ``(Layer, ConvexHull)`` or ``None``; the layer's image stays in HBM until read."""
        prefetch = STATE.mask_prefetch
        if prefetch is not None:  # Scene.render rendered every mask it will need in one batch (scene.py)
            hit = prefetch.get(self, transform, fill_rule, viewport)
            if hit is not prefetch.MISS:
                return hit
        res = self._single_batch(transform, fill_rule, viewport)
        if res is None:
            return None
        ctx, batch, bb = res
        rows, cols = int(bb[2]), int(bb[3])
        buf = ctx.alloc(rows * cols * 8)
        batch.render(buf, _abi.OUT_MASK_F64)
        offset = _offset(bb, viewport)
        layer = Layer._from_device(buf, (rows, cols, 1), offset, pre_alpha=True, linear_rgb=True)
        return layer, ConvexHull(_source=lambda: batch.all_edges()[0])

    def fill(self, transform: Transform, paint, fill_rule: str | None = None, viewport=None, linear_rgb: bool = True):
        """Render path by fill-ing it, S:995-1103: solid colours, gradients, patterns."""
        if paint is None:
            return None
        if isinstance(paint, np.ndarray) and paint.shape == (4,):
            paint = solid_paint(paint, linear_rgb)
            plans = STATE.fill_plans
            if plans:  # (a retained render keeps the plan for the next one: scene._Retained)
                fkey = fill_plan_key(self, transform, fill_rule, paint, viewport)
                res = plans.get(fkey, _NO_PLAN) if STATE.fill_plans_keep else plans.pop(fkey, _NO_PLAN)
            else:
                res = _NO_PLAN
            if res is _NO_PLAN:  # (else: built and planned by Scene.render's pre-pass together with the document's other batches)
                res = self._single_batch(transform, fill_rule, viewport, paint)
            if res is None:
                return None
            ctx, batch, bb = res
            rows, cols = int(bb[2]), int(bb[3])
            if isinstance(batch, FillView):   # (one of the document's node-by-node fills: they share a batch and ONE launch per render)
                buf = batch.layer_buffer()
            else:
                buf = ctx.alloc(rows * cols * 32)
                batch.render(buf, _abi.OUT_FILL_F64)
            layer = Layer._from_device(buf, (rows, cols, 4), _offset(bb, viewport), pre_alpha=True, linear_rgb=linear_rgb)
            return layer, ConvexHull(_source=lambda: batch.all_edges()[0])
        from .paint import gradient_fill, is_gradient  # noqa: PLC0415

        if is_gradient(paint):
            res = self.mask(transform, fill_rule, viewport)
            if res is None:
                return None
            mask, hull = res
            return gradient_fill(paint, mask, hull, transform, linear_rgb), hull
        from .paint import Pattern, pattern_fill  # noqa: PLC0415

        if isinstance(paint, Pattern):
            res = self.mask(transform, fill_rule, viewport)
            if res is None:
                return None
            mask, hull = res
            layer = pattern_fill(paint, mask, hull, transform, linear_rgb)
            return None if layer is None else (layer, hull)
        warnings.warn(f"fill method is not implemented: {paint}")
        return None

    # -- convenience ------------------------------------------------------------------------
    def stroke(self, width: float, linecap: str | None = None, linejoin: str | None = None) -> "Path":
        """Convert the path to its stroke outline, a fill path (Path.stroke, S:1105-1180).  The work is done by
        the native stroker (csrc/svgr_stroke.cpp); quadratic and arc segments are turned into cubics first with the
        same conversions the reference applies (S:1133-1140)."""
        caps = {None: 0, STROKE_CAP_BUTT: 0, STROKE_CAP_ROUND: 1, STROKE_CAP_SQUARE: 2}
        joins = {None: 0, STROKE_JOIN_MITER: 0, STROKE_JOIN_ROUND: 1, STROKE_JOIN_BEVEL: 2}
        if linecap not in caps:
            raise ValueError(f"unkown line cap type: `{linecap}`")
        if linejoin not in joins:
            raise ValueError(f"unknown line join type: `{linejoin}`")
        types, params, sizes = [], [], []
        for sub in self.subpaths:
            if not sub:
                continue
            n0 = len(types)
            for seg in sub:
                t = seg[0]
                if t in PATH_LINES:
                    types.append(t)
                    params.append(np.concatenate([np.asarray(seg[1], dtype=FLOAT).reshape(4), np.zeros(4)]))
                elif t == PATH_CUBIC:
                    types.append(PATH_CUBIC)
                    params.append(np.asarray(seg[1], dtype=FLOAT).reshape(8))
                elif t == PATH_QUAD:
                    types.append(PATH_CUBIC)
                    params.append(quad_to_cubic(seg[1]).reshape(8))
                elif t == PATH_ARC:
                    for c in arc_to_cubics(*seg[1]):
                        types.append(PATH_CUBIC)
                        params.append(c.reshape(8))
                else:
                    raise ValueError(f"unsupported path type: `{t}`")
            sizes.append(len(types) - n0)
        if not types:
            return Path([])
        ot, op, osz = _abi.path_stroke(types, np.array(params), sizes, width, caps[linecap], joins[linejoin])
        return Path.from_segments(ot, op, osz)

    @classmethod
    def from_svg(cls, d: str) -> "Path":
        from .pathdata import parse_path_data  # noqa: PLC0415

        return cls(parse_path_data(d))


class MaskPrefetch:
    """Masks of many (path, transform, rule) jobs over ONE viewport, rendered by a single batch
    (``SVGR_OUT_MASKS_F64``) and handed out by ``Path.mask``.  Purely a cache: a job that was not predicted is
    rendered on demand as before."""

    MISS = object()

    def __init__(self, jobs, viewport, retained: "dict | None" = None):
        self.viewport = tuple(int(v) for v in viewport)
        self.table: dict = {}
        if retained is not None and retained.get("viewport") == self.viewport:
            # a later render of the same document (scene._Retained): the job list, the batch and its plan are the first render's;
            # only the masks themselves are rendered again (their layers are handed out and may be consumed)
            todo, batch = retained["todo"], retained["keep"]
            self.n_jobs = len(todo)
            self._paths = [t[1] for t in todo]
            if batch is not None:
                self._fill_table(todo, batch, _abi.Context.get())
            return
        todo, seen = [], set()
        for path, transform, rule in jobs:
            if rule not in _RULES:
                continue  # Path.mask raises for it
            key = (id(path), transform.m6().tobytes(), _RULES[rule])
            if key in seen or len(path.packed()[0]) == 0:
                continue
            seen.add(key)
            todo.append((key, path, transform, rule))
        self.n_jobs = len(todo)
        # the table is keyed by id(path): hold the paths for as long as the table lives, so that the id of a path that
        # was dropped cannot come back as another path's
        self._paths = [t[1] for t in todo]
        if retained is not None:
            retained.clear()
            retained.update(viewport=self.viewport, todo=todo, keep=None)
        if not todo:
            return
        segs, kinds, offs, m6s, rules = [], [], [0], [], []
        for _key, path, transform, rule in todo:
            s, k = path.packed()
            segs.append(s)
            kinds.append(k)
            offs.append(offs[-1] + len(s))
            m6s.append(transform.m6())
            rules.append(_RULES[rule])
        ctx = _abi.Context.get()
        batch = _abi.Batch(ctx, np.concatenate(segs), np.concatenate(kinds), offs, np.array(m6s), rules,
                           np.zeros((len(todo), 4)), viewport=list(self.viewport), flatness=FLATNESS)
        batch.plan()
        bb = batch.bboxes()
        area = np.where((bb[:, 2] > 0) & (bb[:, 3] > 0), bb[:, 2].astype(np.int64) * bb[:, 3], 0)
        if int(area.sum()) * 8 > (1 << 31):  # more than 2 GiB of masks at once: leave them to the on-demand route
            batch.destroy()
            self.n_jobs = 0
            return
        if retained is not None:
            retained["keep"] = batch
        self._fill_table(todo, batch, ctx)

    def _fill_table(self, todo, batch, ctx):
        buf, loffs, bb = batch.render_masks()
        self._keep = (batch, buf)
        base = buf.ptr
        edges_cache: list = []

        def edges_of(i):
            if not edges_cache:
                edges_cache.append(batch.all_edges())  # (unculled: the hull of a mask covers the whole shape, S:993)
            e, ep = edges_cache[0]
            return e[ep == i]

        for i, (key, _path, _transform, _rule) in enumerate(todo):
            rows, cols = int(bb[i, 2]), int(bb[i, 3])
            if rows <= 0 or cols <= 0:
                self.table[key] = None
                continue
            view = ctx.wrap(base + int(loffs[i]) * 8, rows * cols * 8)
            view._parent = buf  # the view does not own the memory
            layer = Layer._from_device(view, (rows, cols, 1), _offset(bb[i], self.viewport), pre_alpha=True, linear_rgb=True)
            self.table[key] = (layer, ConvexHull(_source=lambda i=i: edges_of(i)))

    def get(self, path, transform, fill_rule, viewport):
        if viewport is None or fill_rule not in _RULES or tuple(int(v) for v in viewport) != self.viewport:
            return self.MISS
        return self.table.get((id(path), transform.m6().tobytes(), _RULES[fill_rule]), self.MISS)


# (the running render's MaskPrefetch and its pre-planned single-path fills -- key -> (ctx, batch, bbox) or None (nothing to draw),
#  consumed by Path.fill -- live in _state.STATE, per thread: STATE.mask_prefetch, STATE.fill_plans, STATE.fill_plans_keep)
_NO_PLAN = object()


def fill_plan_key(path, transform, fill_rule, paint4, viewport):
    return (id(path), transform.key(), fill_rule, paint4.tobytes(), None if viewport is None else tuple(int(v) for v in viewport))


# (STATE.serial numbers the top-level Scene.render calls (scene.py): the shared fills are drawn once per render)
_SHARE_FILLS = __import__("os").environ.get("SVGR_NO_SHARED_FILLS") is None


class _FillSet:
    """The solid fills a document draws node by node (children of filter / mask / bbox-clip nodes), in ONE batch: one plan,
    and per render one geometry pass and one tile launch that writes every fill's layer (SVGR_OUT_FILLS_F64)."""

    __slots__ = ("batch", "refs", "serial", "buf", "offs", "_edges")

    def __init__(self, batch, refs):
        self.batch, self.refs, self.serial, self.buf, self.offs, self._edges = batch, refs, -1, None, None, None

    def draw(self):
        if self.serial != STATE.serial or self.buf is None:
            self.buf, self.offs, _bb = self.batch.render_fills()   # (a fresh buffer: the layers of the last render may still be in use)
            self.serial = STATE.serial
        return self.buf, self.offs

    def all_edges(self):
        if self._edges is None:
            self._edges = self.batch.all_edges()
        return self._edges


class FillView:
    """One fill of a `_FillSet`: what `Path.fill` holds in the place of the fill's own batch."""

    __slots__ = ("fills", "index", "rows", "cols", "dead")

    def __init__(self, fills, index):
        self.fills, self.index, self.rows, self.cols, self.dead = fills, index, 0, 0, False

    def layer_buffer(self):
        buf, offs = self.fills.draw()
        view = buf.ctx.wrap(buf.ptr + int(offs[self.index]) * 32, self.rows * self.cols * 32)
        view._parent = buf   # (the view does not own the memory)
        return view

    def all_edges(self):
        e, ep = self.fills.all_edges()
        mine = ep == self.index
        return e[mine], ep[mine] * 0

    def bboxes(self):
        return self.fills.batch.bboxes()[self.index:self.index + 1]

    def destroy(self):
        if not self.dead:
            self.dead = True
            self.fills.refs -= 1
            if self.fills.refs <= 0:
                self.fills.batch.destroy()


def plan_fills(jobs, viewport):
    """[(path, transform, rule, converted paint)] -> ({key: [ctx, batch or FillView, None]}, [batches to plan]): what
    `Path.fill` would build, unplanned -- one batch for all of them (two or more), each fill a `FillView` of it.
    `finish_fill_plans` turns the entries into what `_single_batch` returns."""
    plans, batches = {}, []
    ctx = _abi.Context.get()
    vp = None if viewport is None else [int(v) for v in viewport]
    todo = []
    for path, transform, rule, paint4 in jobs:
        key = fill_plan_key(path, transform, rule, paint4, viewport)
        if key in plans or rule not in _RULES:
            continue
        segs, kinds = path.packed()
        if len(segs) == 0:
            continue
        plans[key] = None
        todo.append((key, segs, kinds, transform.m6(), _RULES[rule], paint4))
    if len(todo) >= 2 and _SHARE_FILLS and vp is not None:
        offs = [0]
        for t in todo:
            offs.append(offs[-1] + len(t[1]))
        batch = _abi.Batch(ctx, np.concatenate([t[1] for t in todo]), np.concatenate([t[2] for t in todo]), offs,
                           np.array([t[3] for t in todo]), [t[4] for t in todo], np.array([t[5] for t in todo]), viewport=vp, flatness=FLATNESS)
        fills = _FillSet(batch, len(todo))
        for i, t in enumerate(todo):
            plans[t[0]] = [ctx, FillView(fills, i), None]
        batches.append(batch)
    else:
        for key, segs, kinds, m6, rule, paint4 in todo:
            batch = _abi.Batch(ctx, segs, kinds, [0, len(segs)], m6, [rule], [paint4], viewport=vp, flatness=FLATNESS)
            plans[key] = [ctx, batch, None]
            batches.append(batch)
    return plans, batches


def finish_fill_plans(plans):
    shared_bb, too_big = {}, set()
    for key, entry in list(plans.items()):
        ctx, batch, _ = entry
        if isinstance(batch, FillView):
            fs = batch.fills
            if id(fs) not in shared_bb:
                bb_all = fs.batch.bboxes()
                shared_bb[id(fs)] = bb_all
                area = np.where((bb_all[:, 2] > 0) & (bb_all[:, 3] > 0), bb_all[:, 2].astype(np.int64) * bb_all[:, 3], 0)
                if int(area.sum()) * 32 > (1 << 32):   # more than 4 GiB of fill layers at once: each fill for itself, on demand
                    too_big.add(id(fs))
            if id(fs) in too_big:
                batch.destroy()
                del plans[key]
                continue
            bb = shared_bb[id(fs)][batch.index]
        else:
            bb = batch.bboxes()[0]
        if bb[2] <= 0 or bb[3] <= 0:
            batch.destroy()
            plans[key] = None
        else:
            if isinstance(batch, FillView):
                batch.rows, batch.cols = int(bb[2]), int(bb[3])
            plans[key] = (ctx, batch, bb)
    return plans


def _offset(bb, viewport):
    """Layer.offset: np.int64 when unclipped, Python int when viewport-clipped."""
    if viewport is None:
        return (np.int64(bb[0]), np.int64(bb[1]))
    return (int(bb[0]), int(bb[1]))


_PAINT_MEMO: dict = {}  # (4 doubles as bytes, linear_rgb) -> converted paint: documents reuse a handful of colours


def solid_paint(paint: np.ndarray, linear_rgb: bool) -> np.ndarray:
    """The 4-vector colour step of Path.fill (S:1014-1018): premultiplied linear RGBA ->
    premultiplied RGBA of the compositing space.  Four numbers, done on the host in double."""
    # (the memo's arrays are handed out as they are, read-only: a document's thousand fills share a handful of colours, and a copy
    #  per fill was a third of this function)
    if type(paint) is np.ndarray and paint.dtype == FLOAT:
        key = (paint.tobytes(), linear_rgb is True or bool(linear_rgb))
        hit = _PAINT_MEMO.get(key)
        if hit is not None:
            return hit
    out = np.array(paint, dtype=FLOAT)
    key = (out.tobytes(), bool(linear_rgb))
    hit = _PAINT_MEMO.get(key)
    if hit is not None:
        return hit
    if len(_PAINT_MEMO) > 8192:
        _PAINT_MEMO.clear()
    out = _solid_paint(out, linear_rgb)
    out.flags.writeable = False
    _PAINT_MEMO[key] = out
    return out


def _solid_paint(out: np.ndarray, linear_rgb: bool) -> np.ndarray:
    if not linear_rgb:
        rgb, alpha = out[:3], out[3:]
        np.divide(rgb, alpha, out=rgb, where=alpha > 0.0001)
        np.clip(out, 0, 1, out=out)
        small = rgb <= 0.0031308
        rgb[small] = rgb[small] * 12.92
        large = ~small
        rgb[large] = 1.055 * np.power(rgb[large], 1.0 / 2.4) - 0.055
        rgb *= alpha
    return out
