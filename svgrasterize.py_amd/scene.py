"""Scene graph + renderer: the caller of the hot path (reference S:576-859).

Same node types, constructors and ``render`` signature as the reference.  ``render`` has two
routes, both of them HIP:

* batched: a maximal run of solid-colour leaves is handed to the device as ONE paint-ordered batch
  (``svgr_batch_render``) that flattens, bins and composites all of them in a single tile kernel
  (per-path launches cannot win).  A run may contain, besides plain FILL / STROKE leaves under
  GROUP / TRANSFORM nodes: an OPACITY directly over a leaf (folded into the paint), a CLIP whose clip and
  target are single paths (clip source + clipped entry), and a CLIP or an OPACITY over a GROUP of plain
  leaves (an isolated group: its members composite into a group tile on the device, which is clipped /
  faded as a whole, ``svgr_batch_set_groups``).
* per node: everything else (gradient and pattern fills, filters, masks, isolated groups that are not flat)
  is rendered node by node into device-resident Layers and merged with ``Layer.compose`` exactly as the
  reference does.

STROKE nodes are stroked by the native stroker (`Path.stroke`, csrc/svgr_stroke.cpp) and then treated as fills.
"""
from __future__ import annotations

import gc
import math
import textwrap

import numpy as np

from . import _abi
from ._state import RENDER_LOCK, STATE, next_serial
from .geometry import ConvexHull, Path, Transform, solid_paint, _RULES, FLATNESS
from .layer import COMPOSE_IN, COMPOSE_OVER, Layer
from .paint import _SPREAD, is_gradient, needs_mask   # (paint.py imports nothing of this module)

RENDER_FILL, RENDER_STROKE, RENDER_GROUP, RENDER_OPACITY = 0, 1, 2, 3
RENDER_CLIP, RENDER_MASK, RENDER_TRANSFORM, RENDER_FILTER = 4, 5, 6, 7


class _Retained:
    """What a top-level ``Scene.render`` keeps for the NEXT render of the same (scene, transform, viewport, colour space):
    the leaf analysis of every group child, the jobs of the mask pre-pass with their built and planned batch, the runs' and
    the per-node fills' built and planned batches with everything derived from their bboxes.  A second render of an unchanged
    document is the walk plus launches: no leaf analysis, no batch building, no plan (VERDICT r3 #5a).

    OPT-IN (``SVGR_RENDER_CACHE`` = entries kept, or ``set_render_cache(n)``; default 0): the reference's ``Scene.render``
    keeps nothing between calls (S:649-752), and a document whose arrays are edited in place between two renders is simply drawn
    with the new values.  A caller who turns the cache on promises to treat a rendered scene as a value.  Half of that promise is
    checked anyway: the entry is keyed by identity (it holds the scene, so its id cannot come back as another scene's) and
    GUARDED by content for the paints -- every ``np.ndarray`` reachable from the scene outside its paths (solid colours,
    gradient vectors, stop colours) is listed when the entry is made and their bytes are compared before the entry is used
    again (solid colours stacked into one array, the rest through `svgr_hash_buffers`); on a mismatch the render starts over.
    Path geometry is NOT re-read: a `Path` is a value type here as in the reference (S:899-907: its methods return new paths)
    and has kept a packed copy of its segments since round 1 (`Path.packed`), with or without this cache -- listing the
    ~30 000 segment arrays of material-design.svg costs more than rendering it.  ``SVGR_RENDER_CACHE_TRUST`` skips the check;
    ``clear_render_cache()`` drops the entries.

    Dropping an entry only drops references: a batch frees its device memory when its last user is gone
    (`_abi.Batch`'s finalizer) -- the lazy hulls of layers handed out by earlier renders read their batch's edges on first use,
    possibly after the entry that built the batch has been evicted."""

    __slots__ = ("scene", "leaf_memo", "jobs", "run_plans", "fill_plans", "mask_state", "colours", "others", "ptrs", "sizes", "digest")

    def __init__(self, scene):
        self.scene = scene
        self.leaf_memo = {}
        self.jobs = None
        self.run_plans = {}
        self.fill_plans = {}
        self.mask_state = {}
        self.colours = self.others = self.ptrs = self.sizes = self.digest = None

    def fingerprint(self):
        """List the scene's paint arrays (first call) and take their bytes / hash them."""
        if self.colours is None:
            arrays, seen = [], set()
            _paint_arrays(self.scene, arrays, seen)
            self.colours = [a for a in arrays if a.shape == (4,) and a.dtype == np.float64]
            self.others = [a for a in arrays if not (a.shape == (4,) and a.dtype == np.float64)]
            self.ptrs = np.array([a.__array_interface__["data"][0] for a in self.others], dtype=np.uint64)
            self.sizes = np.array([a.nbytes for a in self.others], dtype=np.int64)
        stacked = np.array(self.colours).tobytes() if self.colours else b""
        return (stacked, _abi.hash_buffers(self.ptrs, self.sizes) if len(self.others) else 0)

    def release(self):
        self.run_plans = {}
        self.fill_plans = {}
        self.mask_state = {}
        self.leaf_memo = {}

    def sweep_on_demand(self, serial):
        """Runs that were built during a render, not by the pre-pass (inside bbox-relative masks, pattern tiles, ...), are kept
        only while their key keeps coming back: leaves that the walk makes afresh in every render never match again, and
        their batches would pile up."""
        dead = [k for k, e in self.run_plans.items() if len(e) > 4 and e[4] != serial]
        for k in dead:
            del self.run_plans[k]


def _paint_arrays(obj, arrays, seen, depth=0):
    """Every C-contiguous ndarray reachable from a scene tree OUTSIDE its paths: solid paints, gradient fields, stop colours."""
    if isinstance(obj, np.ndarray):
        if id(obj) not in seen and obj.flags.c_contiguous and obj.dtype != object:
            seen.add(id(obj))
            arrays.append(obj)
        return
    if isinstance(obj, Path):
        return
    if isinstance(obj, (tuple, list)) and depth < 64:
        for item in obj:
            if not isinstance(item, (int, float, str, bool, type(None))):
                _paint_arrays(item, arrays, seen, depth + 1)
        return
    sub = getattr(obj, "scene", None)      # a Pattern's tile
    if isinstance(sub, tuple):
        _paint_arrays(sub, arrays, seen, depth + 1)


_RETAINED: "dict[tuple, _Retained]" = {}   # (insertion-ordered: the oldest entry goes first)
_RETAINED_LOCK = __import__("threading").Lock()
_RETAINED_MAX = int(__import__("os").environ.get("SVGR_RENDER_CACHE", "0"))   # (opt-in: see _Retained)
_RETAINED_TRUST = __import__("os").environ.get("SVGR_RENDER_CACHE_TRUST") is not None


def set_render_cache(entries: int) -> None:
    """Keep the built and planned batches of the last `entries` (scene, transform, viewport) renders for the next render of
    the same one (0: off, the default -- see `_Retained` for what the caller promises)."""
    global _RETAINED_MAX
    _RETAINED_MAX = max(int(entries), 0)
    if _RETAINED_MAX == 0:
        clear_render_cache()


def clear_render_cache() -> None:
    """Drop what ``Scene.render`` retained between renders (the device buffers go when their last user does)."""
    with _RETAINED_LOCK:
        entries = list(_RETAINED.values())
        _RETAINED.clear()
    for st in entries:
        st.release()


class Scene(tuple):
    __slots__ = []

    def __new__(cls, type, args):
        return tuple.__new__(cls, (type, args))

    # -- constructors (S:604-647) ----------------------------------------------------------
    @classmethod
    def fill(cls, path: Path, paint, fill_rule=None) -> "Scene":
        return cls(RENDER_FILL, (path, paint, fill_rule))

    @classmethod
    def stroke(cls, path: Path, paint, width: float, linecap=None, linejoin=None) -> "Scene":
        return cls(RENDER_STROKE, (path, paint, width, linecap, linejoin))

    @classmethod
    def group(cls, children) -> "Scene":
        children = tuple(children)
        if not children:
            raise ValueError("group have to contain at least one child")
        if len(children) == 1:
            return children[0]
        return cls(RENDER_GROUP, children)

    def opacity(self, opacity: float) -> "Scene":
        if opacity > 0.999:
            return self
        return Scene(RENDER_OPACITY, (self, opacity))

    def clip(self, clip: "Scene", bbox_units: bool = False) -> "Scene":
        return Scene(RENDER_CLIP, (self, clip, bbox_units))

    def mask(self, mask: "Scene", bbox_units: bool = False) -> "Scene":
        return Scene(RENDER_MASK, (self, mask, bbox_units))

    def transform(self, transform: Transform) -> "Scene":
        type, args = self
        if type == RENDER_TRANSFORM:
            target, target_transform = args
            return Scene(RENDER_TRANSFORM, (target, transform @ target_transform))
        return Scene(RENDER_TRANSFORM, (self, transform))

    def filter(self, filter) -> "Scene":
        return Scene(RENDER_FILTER, (self, filter))

    def to_path(self, transform: Transform) -> Path:
        """All leaf outlines of the scene as one path, transforms applied on the host, strokes outlined after their
        transform; decorations (opacity, clip, mask, filter) are looked through (S:754-794, a testing aid)."""
        def outlines(scene, tr):
            kind, args = scene
            if kind == RENDER_FILL:
                yield args[0].transform(tr)
            elif kind == RENDER_STROKE:
                path, _paint, width, linecap, linejoin = args
                yield path.transform(tr).stroke(width, linecap, linejoin)
            elif kind == RENDER_GROUP:
                for child in args:
                    yield from outlines(child, tr)
            elif kind == RENDER_TRANSFORM:
                yield from outlines(args[0], tr @ args[1])
            elif kind in (RENDER_OPACITY, RENDER_CLIP, RENDER_MASK, RENDER_FILTER):
                yield from outlines(args[0], tr)
            else:
                raise ValueError(f"unhandled scene type: {kind}")

        return Path([sub for path in outlines(self, transform) for sub in path.subpaths])

    def __repr__(self) -> str:
        """Indented tree dump in the reference's wording (S:796-859)."""
        pad = "  "

        def colour(paint):
            if isinstance(paint, np.ndarray):  # two hex digits per channel, single digits padded on the right (S:856)
                return "#" + "".join(f"{c:0<2x}" for c in (paint * 255).astype(np.uint8))
            return paint

        def dump(scene, depth, out):
            kind, args = scene
            head = pad * depth
            if kind == RENDER_FILL:
                path, paint, rule = args
                out.append(f"{head}FILL fill_rule:{rule} paint:{colour(paint)}\n{textwrap.indent(repr(path), pad * (depth + 1))}")
            elif kind == RENDER_STROKE:
                path, paint, width, linecap, linejoin = args
                out.append(f"{head}STROKE width:{width} linecap:{linecap} linejoin:{linejoin} paint:{colour(paint)}\n"
                           f"{textwrap.indent(repr(path), pad * (depth + 1))}")
            elif kind == RENDER_GROUP:
                out.append(f"{head}GROUP")
                for child in args:
                    dump(child, depth + 1, out)
            elif kind == RENDER_OPACITY:
                out.append(f"{head}OPACITY {args[1]}")
                dump(args[0], depth + 1, out)
            elif kind in (RENDER_CLIP, RENDER_MASK):
                target, shape, bbox_units = args
                name, shape_name = ("CLIP", "CLIP_PATH") if kind == RENDER_CLIP else ("MASK", "MAKS_PATH")
                out.append(f"{head}{name} bbox_units:{bbox_units}")
                out.append(f"{head}{pad}{shape_name}")
                dump(shape, depth + 2, out)
                out.append(f"{head}{pad}{name}_TARGET")
                dump(target, depth + 2, out)
            elif kind == RENDER_TRANSFORM:
                out.append(f"{head}TRANSFORM {args[1]}")
                dump(args[0], depth + 1, out)
            elif kind == RENDER_FILTER:
                out.append(f"{head}FILTER {args[1]}")
                dump(args[0], depth + 1, out)
            else:
                raise ValueError(f"unhandled scene scene[0]: {kind}")
            return out

        return "\n".join(dump(self, 0, []))

    # -- render (S:649-752) ------------------------------------------------------------------
    def render(self, transform: Transform, mask_only: bool = False, viewport=None, linear_rgb: bool = False):
        """Render graph; returns ``(Layer, ConvexHull)`` or ``None`` (S:649-752).

        The outermost call first renders, in ONE batch, every ``Path.mask`` the per-node route is going to ask for
        (clip paths, gradient-filled paths): hundreds of single-path launches become one.  It also builds the batch of every
        run of fills the walk will meet and plans them all behind one wait (``svgr_batch_plan_many``)."""
        # (one context, one stream, no lock in the library: the renders of a process run one after the other; the lock is re-entrant)
        with RENDER_LOCK:
            # (a render inside a render -- a pattern's tile -- walks without a pre-pass of its own: the outer call's state stays)
            if STATE.leaf_memo is not None or STATE.mask_prefetch is not None or viewport is None or self[0] in (RENDER_FILL, RENDER_STROKE):
                return self._render(transform, mask_only, viewport, linear_rgb)
            return self._render_top(transform, mask_only, viewport, linear_rgb)

    def _render_top(self, transform: Transform, mask_only: bool, viewport, linear_rgb: bool):
        from . import displaylist, geometry  # noqa: PLC0415

        # A scene that is batch entries and nothing else (solid fills / strokes under transforms, single-path clips, opacity and
        # clip groups) is compiled ONCE to flat arrays -- nothing of it depends on the render transform but the leaves' matrices --
        # and drawn from them: no tree walk, no leaf tuples, no per-leaf packing (displaylist.py).  Same batch, same picture.
        if _RETAINED_MAX == 0 and not mask_only and (self[0] == RENDER_GROUP or _NODE_RUNS):
            dl = displaylist.get(self, linear_rgb)
            if dl is not None:
                res = dl.render(transform, viewport, linear_rgb)
                if self[0] != RENDER_GROUP or res is None:
                    return res
                group = Layer.compose([res[0]], COMPOSE_OVER, linear_rgb)   # (what the GROUP's own loop ends in, S:686-688)
                if not group:
                    return None
                return group, ConvexHull.merge([res[1]])

        key = (id(self), transform.key(), tuple(int(v) for v in viewport), bool(linear_rgb), bool(mask_only))
        st = None
        if _RETAINED_MAX > 0:
            with _RETAINED_LOCK:
                st = _RETAINED.pop(key, None)   # (taken out: re-inserted as the youngest when the render succeeds)
        if st is not None and not _RETAINED_TRUST and st.fingerprint() != st.digest:
            # a paint of the document was edited in place since the entry was made: the reference would draw the new values
            st.release()
            st = None
        warm = st is not None
        if st is None:
            st = _Retained(self)
        STATE.leaf_memo = st.leaf_memo
        STATE.retain = st if _RETAINED_MAX > 0 else None
        STATE.serial = next_serial()
        # the walk allocates thousands of short-lived tuples and no cycles: the cyclic collector's generation-0 sweeps find
        # nothing and cost 0.5-2.5 ms of a 17 ms render (profiles/gc_experiment.py).  Pausing it is a side effect on the
        # caller's process, so it is opt-in: SVGR_PAUSE_GC=1 (bench.py's scene workloads set it and say so)
        gc_paused = _PAUSE_GC and gc.isenabled()
        if gc_paused:
            gc.disable()
        ok = False
        try:
            if not warm:
                jobs: list = []
                runs: list = []
                fills: list = []
                _collect_mask_jobs(self, transform, mask_only, linear_rgb, jobs, runs, fills)
                st.jobs = jobs
                if len(runs) + len(fills) >= 2:
                    st.run_plans, st.fill_plans = _plan_runs(runs, fills, viewport, linear_rgb)
            if len(st.jobs) >= 4:
                STATE.mask_prefetch = geometry.MaskPrefetch(st.jobs, viewport, st.mask_state if STATE.retain is not None else None)
            STATE.run_plans, STATE.fill_plans = st.run_plans, st.fill_plans
            STATE.fill_plans_keep = STATE.retain is not None
            _prefetch_windows(st.run_plans)
            res = self._render(transform, mask_only, viewport, linear_rgb)
            ok = True
            return res
        finally:
            serial = STATE.serial
            STATE.mask_prefetch = None
            STATE.leaf_memo = None
            STATE.run_plans = None
            STATE.fill_plans = None
            STATE.fill_plans_keep = False
            STATE.retain = None
            if ok and _RETAINED_MAX > 0:
                st.sweep_on_demand(serial)
                if not warm and not _RETAINED_TRUST:
                    st.digest = st.fingerprint()
                with _RETAINED_LOCK:
                    _RETAINED[key] = st
                    evicted = [_RETAINED.pop(next(iter(_RETAINED))) for _ in range(max(len(_RETAINED) - _RETAINED_MAX, 0))]
                for old_st in evicted:
                    old_st.release()
            else:
                st.release()   # (a failed render keeps nothing; with the cache off: the runs the walk did not come to after all)
            if gc_paused:
                gc.enable()

    def _render(self, transform: Transform, mask_only: bool = False, viewport=None, linear_rgb: bool = False, _asked: bool = False):
        kind, args = self
        # (`_asked`: the caller -- a GROUP's loop -- has asked `_leaves_memo` about this node already: it is not batch entries)
        if _NODE_RUNS and not _asked and kind != RENDER_GROUP and not mask_only and viewport is not None and STATE.leaf_memo is not None:
            # Inside a top-level render, a node that is batch entries and nothing else -- the FILL under a FILTER, an OPACITY over a
            # gradient fill, a CLIP of one path by another -- is a run of its own: a window of the document's shared batch, drawn
            # with all the others, instead of a Path.fill / Path.mask with launches of its own (a GROUP cuts its children into runs
            # itself, below).  `_collect_mask_jobs` routes the same way.
            leaves = _leaves_memo(self, transform, linear_rgb)
            if leaves is not None:
                return _render_run(leaves, viewport, linear_rgb)
        if kind == RENDER_FILL:
            path, paint, fill_rule = args
            if mask_only:
                return path.mask(transform, fill_rule=fill_rule, viewport=viewport)
            return path.fill(transform, paint, fill_rule=fill_rule, viewport=viewport, linear_rgb=linear_rgb)

        if kind == RENDER_STROKE:  # S:666-672: stroke outline (native stroker), then an ordinary fill
            path, paint, width, linecap, linejoin = args
            stroke = path.stroke(width, linecap, linejoin)
            if mask_only:
                return stroke.mask(transform, viewport=viewport)
            return stroke.fill(transform, paint, viewport=viewport, linear_rgb=linear_rgb)

        if kind == RENDER_GROUP:
            layers, hulls = [], []
            run: list = []  # pending batchable leaves: (path, m6, rule, paint4)

            def flush():
                if not run:
                    return
                res = _render_run(run, viewport, linear_rgb)
                run.clear()
                if res is not None:
                    layers.append(res[0])
                    hulls.append(res[1])

            for child in args:
                leaves = None if mask_only else _leaves_memo(child, transform, linear_rgb)
                if leaves is not None:
                    run.extend(leaves)
                    continue
                flush()
                res = child._render(transform, mask_only, viewport, linear_rgb, not mask_only)
                if res is None:
                    continue
                layers.append(res[0])
                hulls.append(res[1])
            flush()
            group = Layer.compose(layers, COMPOSE_OVER, linear_rgb)
            if not group:
                return None
            return group, ConvexHull.merge(hulls)

        if kind == RENDER_OPACITY:
            target, opacity = args
            res = target._render(transform, mask_only, viewport, linear_rgb)
            if res is None:
                return None
            layer, hull = res
            return layer.opacity(opacity, linear_rgb), hull

        if kind == RENDER_CLIP:
            target, clip, bbox_units = args
            res = target._render(transform, mask_only, viewport, linear_rgb)
            if res is None:
                return None
            image, hull = res
            if bbox_units:
                transform = hull.bbox_transform(transform)
            clip_res = clip._render(transform, True, viewport, linear_rgb)
            if clip_res is None:
                return None
            mask, _ = clip_res
            result = Layer.compose([mask, image], COMPOSE_IN, linear_rgb)
            if result is None:
                return None
            return result, hull

        if kind == RENDER_TRANSFORM:
            target, target_transform = args
            return target._render(transform @ target_transform, mask_only, viewport, linear_rgb)

        if kind == RENDER_MASK:  # luminance mask (S:721-741)
            target, mask_scene, bbox_units = args
            res = target._render(transform, mask_only, viewport, linear_rgb)
            if res is None:
                return None
            image, hull = res
            if bbox_units:
                transform = hull.bbox_transform(transform)
            mask_res = mask_scene._render(transform, mask_only, viewport, linear_rgb)
            if mask_res is None:
                return None
            mask = mask_res[0].luminance_mask(linear_rgb)
            result = Layer.compose([mask, image], COMPOSE_IN, linear_rgb)
            if result is None:
                return None
            return result, hull
        if kind == RENDER_FILTER:
            target, flt = args
            res = target._render(transform, mask_only, viewport, linear_rgb)
            if res is None:
                return None
            image, hull = res
            return flt(transform, image), hull
        raise ValueError(f"unhandled scene type: {kind}")

    # -- whole-scene batched render: the bench / production entry -----------------------------
    def leaves(self, transform: Transform, linear_rgb: bool = False):
        """Flatten to paint-ordered solid leaves if the whole scene is batchable, else None."""
        return _batchable_leaves(self, transform, linear_rgb)


def _collect_mask_jobs(scene: Scene, transform: Transform, mask_only: bool, linear_rgb: bool, jobs: list, runs=None, fills=None) -> None:
    """(path, transform, rule) of every Path.mask the per-node route of `render` will call: leaves rendered
    ``mask_only`` (clip subtrees) and gradient-filled leaves.  Mirrors the routing of `_render`; a wrong guess only
    costs an unused mask or an on-demand one.  `runs` (a list) also receives every run of batchable leaves a GROUP will
    flush, in the order of the walk; `fills` the solid fills that go node by node: (path, transform, rule, paint)."""
    kind, args = scene
    if _NODE_RUNS and kind != RENDER_GROUP and not mask_only and runs is not None:
        leaves = _leaves_memo(scene, transform, linear_rgb, store=True)   # (as Scene._render routes it)
        if leaves is not None:
            if leaves:
                runs.append(leaves)
            return
    if kind == RENDER_FILL:
        path, paint, rule = args
        if mask_only or needs_mask(paint):
            jobs.append((path, transform, rule))
        elif fills is not None and isinstance(paint, np.ndarray) and paint.shape == (4,):
            fills.append((path, transform, rule, paint))
    elif kind == RENDER_STROKE:
        if mask_only or needs_mask(args[1]):
            jobs.append((_stroked(scene), transform, None))
        elif fills is not None and isinstance(args[1], np.ndarray) and args[1].shape == (4,):
            fills.append((_stroked(scene), transform, None, args[1]))
    elif kind == RENDER_GROUP:
        run: list = []
        for child in args:
            leaves = None if mask_only else _leaves_memo(child, transform, linear_rgb, store=True)
            if leaves is not None:
                run.extend(leaves)  # goes into a solid-fill batch
                continue
            if run and runs is not None:
                runs.append(run)
            run = []
            _collect_mask_jobs(child, transform, mask_only, linear_rgb, jobs, runs, fills)
        if run and runs is not None:
            runs.append(run)
    elif kind == RENDER_TRANSFORM:
        _collect_mask_jobs(args[0], transform @ args[1], mask_only, linear_rgb, jobs, runs, fills)
    elif kind in (RENDER_OPACITY, RENDER_FILTER):
        _collect_mask_jobs(args[0], transform, mask_only, linear_rgb, jobs, runs, fills)
    elif kind == RENDER_CLIP:
        target, clip, bbox_units = args
        _collect_mask_jobs(target, transform, mask_only, linear_rgb, jobs, runs, fills)
        if not bbox_units:  # (objectBoundingBox clips get their transform from the target's hull: on demand)
            _collect_mask_jobs(clip, transform, True, linear_rgb, jobs, runs, fills)
    elif kind == RENDER_MASK:
        _collect_mask_jobs(args[0], transform, mask_only, linear_rgb, jobs, runs, fills)


# (STATE.run_plans: during one top-level render, run key -> [leaves, planned batch, ...] from the pre-pass; STATE.serial numbers the
#  top-level renders: a shared batch runs its geometry once per render)

# The runs of a document share ONE device batch (the filter nodes cut icons.svg's paint order into dozens of
# runs, each a batch of its own with its own five geometry launches and its own plan).  Every run gets a range of rows of a
# tall canvas to itself -- its leaves' matrices are moved down by a whole number of bands --, the geometry kernels run once
# for all of them, and a run's layer is a render window of its range (svgr_batch_render_window, SVGR_RENDER_SAME_GEOMETRY).
# A run whose geometry reaches far beyond the viewport's rows keeps a batch of its own (the range would be mostly air).
_MERGE_RUNS = __import__("os").environ.get("SVGR_NO_MERGED_RUNS") is None
_MERGE_MAX_TILES = 1 << 19     # tiles of the tall canvas (its per-tile tables grow with them); what does not fit starts another one
_ROW_PAD = 4                   # rows kept free around a run's geometry (the anti-aliasing reaches one pixel)
MERGE_STATS = {"shared_batches": 0, "runs_sharing": 0, "runs_alone": 0}   # counted since import (tests, profiles)


class _SharedBatch:
    """A batch several runs draw from."""

    __slots__ = ("batch", "refs", "serial", "_bboxes", "_edges", "shifts", "vrows", "_view_boxes", "leaves", "starts", "_windows")

    def __init__(self, batch, refs, shifts=None, vrows=None, leaves=None, starts=None):
        self.batch, self.refs, self.serial, self._bboxes, self._edges = batch, refs, -1, None, None
        self.shifts, self.vrows, self._view_boxes = shifts, vrows, None   # per leaf: the rows it was moved down by; the viewport's rows
        self.leaves, self.starts, self._windows = leaves, starts, None    # all the runs' leaves, the first leaf of every run

    def run_windows(self):
        """`_run_window` of every run at once: (first leaf of a run -> its window or None, hull membership of every leaf).  The runs
        are independent stretches of one leaf list (a clip source stands in front of what it clips, inside its run), so the leaves'
        effective boxes are one computation and a run's window the reduction of its stretch."""
        if self._windows is None:
            n = len(self.leaves)
            painted, box, ok = _effective_boxes(self.leaves, self.view_boxes())
            big = np.iinfo(np.int64).max
            lo = np.full((n, 2), big, dtype=np.int64)
            hi = np.full((n, 2), -big, dtype=np.int64)
            idx = painted[ok]
            lo[idx] = box[ok][:, :2]
            hi[idx] = box[ok][:, 2:]
            starts = np.asarray(self.starts, dtype=np.int64)
            rlo = np.minimum.reduceat(lo, starts, axis=0)
            rhi = np.maximum.reduceat(hi, starts, axis=0)
            in_hull = np.zeros(n, dtype=bool)
            in_hull[painted] = ok
            wins = {}
            for k, first in enumerate(self.starts):
                r0, c0, r1, c1 = int(rlo[k, 0]), int(rlo[k, 1]), int(rhi[k, 0]), int(rhi[k, 1])
                wins[first] = None if r0 == big else (r0, c0, r1 - r0, c1 - c0)
            self._windows = (wins, in_hull)
        return self._windows

    def bboxes(self):
        if self._bboxes is None:
            self._bboxes = self.batch.bboxes()
        return self._bboxes

    def view_boxes(self):
        """Every leaf's bbox in the viewport's own rows, clipped to them (the shared canvas is taller than the viewport): what
        `_RunView.bboxes` hands out a slice of, computed for all runs at once."""
        if self._view_boxes is None:
            bb = self.bboxes().astype(np.int64)
            r0 = bb[:, 0] - self.shifts
            r1 = r0 + np.maximum(bb[:, 2], 0)
            v0, v1 = self.vrows
            c0, c1 = np.maximum(r0, v0), np.minimum(r1, v1)
            out = bb.copy()
            out[:, 0] = c0
            out[:, 2] = np.where((bb[:, 2] > 0) & (c1 > c0), c1 - c0, 0)
            self._view_boxes = out
        return self._view_boxes

    def all_edges(self):
        if self._edges is None:
            self._edges = self.batch.all_edges()
        return self._edges


class _RunView:
    """A run's part of a shared batch: the paths [lo, hi) and the rows its geometry was moved down by.  Quacks like the
    batch `_render_run` used to hold for the run alone."""

    __slots__ = ("shared", "lo", "hi", "shift", "vrows", "dead", "ready")

    def __init__(self, shared, lo, hi, shift, vrows):
        self.shared, self.lo, self.hi, self.shift, self.vrows, self.dead = shared, lo, hi, shift, vrows, False
        self.ready = None   # (render serial, buffer): the run's window as `_prefetch_windows` drew it for this render

    def bboxes(self):
        """The run's bboxes in the viewport's own rows, clipped to them (the shared canvas is taller than the viewport)."""
        return self.shared.view_boxes()[self.lo:self.hi]

    def all_edges(self):
        edges, edge_path = self.shared.all_edges()
        mine = (edge_path >= self.lo) & (edge_path < self.hi)
        e = edges[mine].copy()
        e[:, :, 0] -= self.shift
        return e, edge_path[mine] - self.lo

    def render(self, out, kind, flags=0, window=None):
        sh = self.shared
        if sh.serial == STATE.serial:
            flags |= _abi.RENDER_SAME_GEOMETRY   # (another run of the same render drew from this batch already)
        sh.serial = STATE.serial
        r0, c0, rows, cols = window
        sh.batch.render(out, kind, flags, window=(r0 + self.shift, c0, rows, cols))

    def destroy(self):
        if not self.dead:
            self.dead = True
            self.shared.refs -= 1
            if self.shared.refs <= 0:
                self.shared.batch.destroy()


def _row_extent(leaf):
    """Rows (first device coordinate) the control points of a leaf's path can reach: the corners of the path's box in user
    space under the leaf's matrix (a superset).  None: not finite."""
    path, m6 = leaf[0], leaf[1]
    box = path.user_box()
    if box is None:
        return None
    x0, y0, x1, y1 = box
    a, b, t = float(m6[0]), float(m6[1]), float(m6[2])
    r = (a * x0 + b * y0 + t, a * x1 + b * y0 + t, a * x0 + b * y1 + t, a * x1 + b * y1 + t)
    lo, hi = min(r), max(r)
    if not (lo == lo and hi == hi) or hi - lo > 1e9 or abs(lo) > 1e9:
        return None
    return lo, hi


def _row_extents(leaves):
    """`_row_extent` of every leaf of a run as two arrays (lowest, highest row), or None when some leaf has none."""
    boxes = [leaf[0].user_box() for leaf in leaves]
    if None in boxes:
        return None
    n = len(leaves)
    box = np.array(boxes, dtype=np.float64).reshape(n, 4)
    m = np.concatenate([leaf[1] for leaf in leaves]).reshape(n, 6)
    a, b, t = m[:, 0], m[:, 1], m[:, 2]
    x0, y0, x1, y1 = box[:, 0], box[:, 1], box[:, 2], box[:, 3]
    r = np.stack([a * x0 + b * y0 + t, a * x1 + b * y0 + t, a * x0 + b * y1 + t, a * x1 + b * y1 + t])
    lo, hi = r.min(axis=0), r.max(axis=0)
    # (NaN fails every comparison; the limits as in `_row_extent`)
    if not bool(np.all((lo == lo) & (hi == hi) & (hi - lo <= 1e9) & (np.abs(lo) <= 1e9))):
        return None
    return lo, hi


def _shift_leaf(leaf, shift):
    """The leaf moved down by `shift` rows (a whole number of bands)."""
    if shift == 0:
        return leaf
    path, m6, rule, paint4, flags, group, grad = leaf
    m6 = np.array(m6, dtype=np.float64)
    m6[2] += shift
    if grad is not None:
        # the gradient's frame moves with the path: user = A (row - shift, col) + t, i.e. the same matrix with t - A[:, 0] * shift
        # (only the device -> user matrix of the description depends on the render transform: paint._common)
        g0, keep, paint, transform, lin = grad
        g = type(g0).from_buffer_copy(g0)
        u = g.user_m6
        u[2] -= u[0] * shift
        u[5] -= u[3] * shift
        grad = (g, keep, paint, transform, lin)
    return (path, m6, rule, paint4, flags, group, grad)


def _merge_runs(todo, viewport):
    """[(key, leaves)] -> {key: [leaves, batch or _RunView]} with the runs that can share a batch sharing one (per
    _MERGE_MAX_ROWS of canvas), the others alone.  Returns (plans, batches to plan)."""
    plans, batches = {}, []
    v0, vrows = int(viewport[0]), int(viewport[2])
    v1 = v0 + vrows
    over = max(vrows, 2048)      # how far beyond the viewport's rows a run's geometry may reach and still share
    tr = _abi.tile_rows()
    max_rows = max(_MERGE_MAX_TILES // max(-(-int(viewport[3]) // _abi.tile_cols()), 1), 1) * tr
    packs, cur, base = [], [], 0
    for key, leaves in todo:
        lo, hi = float(v0), float(v1)
        ok = _MERGE_RUNS and len(todo) > 1
        if ok:
            if len(leaves) <= 6:   # (a run of a few leaves: the arrays cost more than the leaves)
                exts = [_row_extent(leaf) for leaf in leaves]
                ext = None if None in exts else (np.array([e[0] for e in exts]), np.array([e[1] for e in exts]))
            else:
                ext = _row_extents(leaves)
            if ext is None:
                ok = False
            else:
                e_lo, e_hi = ext
                # (a two-circle gradient asks "any pixel of the fill's LAYER with det < 0" (S:1627), and the layer is the bbox
                #  clipped to the batch's viewport: such a fill shares only when it lies inside the viewport's rows anyway)
                for i, leaf in enumerate(leaves):
                    if leaf[6] is not None and _is_focal(leaf[6]) and (e_lo[i] < v0 + 1 or e_hi[i] > v1 - 1):
                        ok = False
                        break
                lo, hi = min(lo, float(e_lo.min())), max(hi, float(e_hi.max()))
            ok = ok and v0 - lo <= over and hi - v1 <= over
        if not ok:
            try:
                batch = build_batch(leaves, viewport)
            except Exception:  # noqa: BLE001
                continue
            plans[key] = [leaves, batch]
            batches.append(batch)
            MERGE_STATS["runs_alone"] += 1
            continue
        lo_b = (int(np.floor(lo)) - _ROW_PAD) // tr * tr          # the run's range starts at a band border ...
        height = -(-(int(np.ceil(hi)) + _ROW_PAD - lo_b) // tr) * tr
        if cur and base + height > max_rows:
            packs.append((cur, base))
            cur, base = [], 0
        cur.append((key, leaves, base - lo_b))                    # ... and its rows move down by a whole number of bands
        base += height
    if cur:
        packs.append((cur, base))
    for members, total in packs:
        if len(members) == 1:
            key, leaves, _shift = members[0]
            try:
                batch = build_batch(leaves, viewport)
            except Exception:  # noqa: BLE001
                continue
            plans[key] = [leaves, batch]
            batches.append(batch)
            continue
        merged, spans, shifts = [], [], []
        for key, leaves, shift in members:
            spans.append((key, leaves, len(merged), len(merged) + len(leaves), shift))
            merged.extend(leaves)                # (moved down by `shift` rows inside build_batch: one addition per column, not a leaf each)
            shifts.extend([shift] * len(leaves))
        try:
            batch = build_batch(merged, [0, int(viewport[1]), total, int(viewport[3])], row_shift=shifts)
        except Exception:  # noqa: BLE001  (the runs then plan for themselves, on demand)
            continue
        shared = _SharedBatch(batch, len(spans), np.asarray(shifts, dtype=np.int64), (v0, v1), merged, [sp[2] for sp in spans])
        MERGE_STATS["shared_batches"] += 1
        MERGE_STATS["runs_sharing"] += len(spans)
        for key, leaves, lo_p, hi_p, shift in spans:
            plans[key] = [leaves, _RunView(shared, lo_p, hi_p, shift, (v0, v1))]
        batches.append(batch)
    return plans, batches



def _run_key(run, viewport):
    """Identity of a run of leaves: the leaf tuples come out of the per-render memo, the same objects in both walks.  EVERY
    leaf counts (two runs of shared sub-scenes may agree in their first and last leaf and their length), and so does the
    viewport the batch was built for (a render inside the render -- a pattern's tile -- has another)."""
    return (tuple(map(id, run)), tuple(int(v) for v in viewport))


# The pre-pass plans at most this many batches up front (each holds its own work buffers until it is drawn: without a cap the
# device memory in flight grows with the number of fills of the document); the rest plan on demand, as before the pre-pass.
_PREPLAN_MAX = 192


def _plan_runs(runs, fills, viewport, linear_rgb):
    """Build the batch of every run and of every per-node solid fill and plan them all behind one wait.  A batch that fails
    to build is left to `_render_run` / `Path.fill` (which then raise where the reference would).
    Returns (run plans, fill plans)."""
    from . import geometry  # noqa: PLC0415

    todo, seen = [], set()
    for run in runs:
        if len(todo) >= _PREPLAN_MAX:
            break
        key = _run_key(run, viewport)
        if key in seen:
            continue
        leaves = _drop_empty(run)
        if not leaves:
            continue
        seen.add(key)
        todo.append((key, leaves))
    plans, batches = _merge_runs(todo, viewport)
    try:
        fill_plans, fill_batches = geometry.plan_fills(
            [(p, t, r, geometry.solid_paint(c, linear_rgb)) for p, t, r, c in fills[: max(_PREPLAN_MAX - len(batches), 0)]], viewport)
    except Exception:  # noqa: BLE001
        fill_plans, fill_batches = {}, []
    try:
        _abi.Batch.plan_many(batches + fill_batches)
        for b in batches:
            _resolve_frames(b)
        geometry.finish_fill_plans(fill_plans)
    except Exception:  # noqa: BLE001  (one bad batch: let every run / fill plan for itself and report its own error)
        for b in batches + fill_batches:
            b.destroy()
        return {}, {}   # (views of a destroyed shared batch are dropped with the dict)
    return plans, fill_plans




def _leaves_memo(child: Scene, transform: Transform, linear_rgb: bool, store: bool = False):
    memo = STATE.leaf_memo
    if memo is None:
        return _batchable_leaves(child, transform, linear_rgb)
    key = (id(child), transform.key(), linear_rgb)
    if key in memo:
        return memo[key]
    # (only GROUP nodes have a memo of their own in front of the analysis: `_batchable_leaves`)
    res = _batchable_leaves(child, transform, linear_rgb) if child[0] == RENDER_GROUP else _batchable_leaves_(child, transform, linear_rgb)
    if store:
        memo[key] = res
    return res


_STROKE_CACHE: dict = {}


def _stroked(scene: Scene) -> Path:
    """Outline of a STROKE node, computed once per node object (strokes do not depend on the transform, S:668)."""
    key = id(scene)
    hit = _STROKE_CACHE.get(key)
    if hit is not None and hit[0] is scene:
        return hit[1]
    path, _paint, width, linecap, linejoin = scene[1]
    out = path.stroke(width, linecap, linejoin)
    if len(_STROKE_CACHE) > 4096:
        _STROKE_CACHE.clear()
    _STROKE_CACHE[key] = (scene, out)
    return out


_BATCH_GROUPS = __import__("os").environ.get("SVGR_NO_BATCH_GROUPS") is None  # (off: isolated groups take the per-node route)
_BATCH_GRADS = __import__("os").environ.get("SVGR_NO_BATCH_GRADIENTS") is None  # (off: gradient fills take the per-node route)
_ONES = np.ones(4)
_ZERO4 = np.zeros(4)
_ZERO4.flags.writeable = False   # (the paint of every clip source)
_PAUSE_GC = __import__("os").environ.get("SVGR_PAUSE_GC") is not None  # (set: the cyclic collector pauses for the duration of a top-level Scene.render)
_GRAD_ABI_MEMO: dict = {}  # (id(gradient), transform bytes, linear_rgb) -> (gradient, svgr_gradient struct, keep-alive)


def _leaf(path, m6, rule, paint4, flags=0, group=None, grad=None):
    """One batch entry: (path, m6, rule, paint4, flags, group, grad).  flags: 0 painted, 1 clip source, 2 clipped by the
    clip source in front of it; group: tag of the isolated group it is a member of; grad: (Gradient struct, keep-alive) of a
    gradient paint -- paint4 is then the multiplier of the evaluated colour (ones, or an opacity)."""
    return (path, m6, rule, paint4, flags, group, grad)


def _gradient_leaf(path, paint, rule, transform: Transform, linear_rgb: bool, opacity):
    """The batch entry of a gradient-filled path, or None when the fill has to go node by node: objectBoundingBox units
    (the frame comes from the path's hull), a colour space of its own (the fill layer is converted when composed), more
    stops than the device block carries.  Path.fill's gradient branch, S:1021-1047."""

    if not _BATCH_GRADS or not 1 <= len(paint.stops) <= 32:
        return None
    if paint.linear_rgb is not None and bool(paint.linear_rgb) != bool(linear_rgb):
        return None
    if paint.spread not in _SPREAD:
        raise ValueError(f"invalid spread method: {paint.spread}")
    if paint.bbox_units:
        # objectBoundingBox units: the gradient's frame is the bounding box of the path's hull (S:1023-1027), known once the batch's
        # plan has flattened the path (`_resolve_frames`: svgr_batch_get_extents, then the descriptions once more).  Only under a
        # transform that keeps the axes apart is the hull's box in user space the box of its extreme device coordinates.
        if not _BATCH_BBOX_GRADS or not _axes_apart(transform):
            return None
        mult = _ONES if opacity is None else _ONES * opacity
        return _leaf(path, transform.m6(), _RULES[rule], mult, grad=(None, None, paint, transform, bool(linear_rgb)))
    # the ABI description of (this gradient, this transform) is a pure function of both: packed once (the struct is copied
    # by Batch.set_gradients, never written to)
    key = (id(paint), transform.key(), bool(linear_rgb))
    hit = _GRAD_ABI_MEMO.get(key)
    if hit is not None and hit[0] is paint:
        g, keep = hit[1], hit[2]
    else:
        g, keep = paint.abi(transform.invert, linear_rgb)
        if len(_GRAD_ABI_MEMO) > 8192:
            _GRAD_ABI_MEMO.clear()
        _GRAD_ABI_MEMO[key] = (paint, g, keep)
    mult = _ONES if opacity is None else _ONES * opacity  # Layer.opacity over the leaf: image * opacity (S:174)
    return _leaf(path, transform.m6(), _RULES[rule], mult, grad=(g, keep, paint, transform, bool(linear_rgb)))


_NODE_RUNS = __import__("os").environ.get("SVGR_NO_NODE_RUNS") is None  # (off: a batchable node outside a GROUP's children goes node by node)
_BATCH_BBOX_GRADS = __import__("os").environ.get("SVGR_NO_BATCH_BBOX_GRADIENTS") is None  # (off: objectBoundingBox gradients go node by node)
_AXES_MEMO: dict = {}


def _axes_apart(transform: Transform) -> bool:
    """The inverse of `transform` takes rows and columns to one user axis each (scales, translations, an x / y swap: every
    entry it multiplies the other coordinate by is exactly zero).  Then `transform.invert(points)` is monotone per axis, rounding
    included, and the hull's bounding box in user space (ConvexHull.bbox, S:2010-2016) is the box of the four corners of its
    device-space extent."""
    key = transform.key()
    hit = _AXES_MEMO.get(key)
    if hit is None:
        inv = np.asarray(transform.invert.m, dtype=np.float64)
        a, b, c, d = float(inv[0, 0]), float(inv[0, 1]), float(inv[1, 0]), float(inv[1, 1])
        ok = all(math.isfinite(v) for v in (a, b, c, d, float(inv[0, 2]), float(inv[1, 2])))
        hit = ok and ((b == 0.0 and c == 0.0 and a != 0.0 and d != 0.0) or (a == 0.0 and d == 0.0 and b != 0.0 and c != 0.0))
        if len(_AXES_MEMO) > 8192:
            _AXES_MEMO.clear()
        _AXES_MEMO[key] = hit
    return hit


def _is_focal(grad) -> bool:
    """A two-circle radial gradient (the description's kind 3; for a frame still pending, the paint says it)."""
    if grad[0] is not None:
        return grad[0].kind == 3
    paint = grad[2]
    return hasattr(paint, "fcenter") and not (paint.fcenter is None and paint.fradius is None)


def _resolve_frames(batch) -> None:
    """The objectBoundingBox gradients of a planned batch get their frames: hull.bbox_transform(transform) (S:1023-1027) from the
    extent of the path's flattened points (svgr_batch_get_extents), the descriptions set once more (same assignment: the plan and
    its geometry pass stay).  Between the plan and the batch's first render."""
    fr = getattr(batch, "_frames", None)
    if not fr:
        return
    batch._frames = None
    path_grad, grads, pending = fr
    ext = batch.extents()
    keep = []
    for gi, pi, paint, transform, lin, shift in pending:
        r0, c0, r1, c1 = (float(v) for v in ext[pi])
        if not all(math.isfinite(v) for v in (r0, c0, r1, c1)):
            continue   # (no edge: nothing of the path is drawn)
        if shift:      # (the run's rows inside a shared canvas: the frame is that of the document's own rows, moved like the path)
            r0, r1 = r0 - shift, r1 - shift
        pts = transform.invert(np.array([[r0, c0], [r0, c1], [r1, c0], [r1, c1]], dtype=np.float64))
        min_x, min_y = pts.min(axis=0)
        max_x, max_y = pts.max(axis=0)
        w, h = max_x - min_x, max_y - min_y
        frame = transform if (w <= 0 and h <= 0) else transform.translate(min_x, min_y).scale(w, h)   # ConvexHull.bbox_transform
        g, k = paint.abi(frame.invert, lin)
        if shift:
            u = g.user_m6
            u[2] -= u[0] * shift
            u[5] -= u[3] * shift
        grads[gi] = g
        keep.append(k)
    batch.set_gradients(path_grad, grads)
    del keep


def _new_group(opacity: float, clipped: bool):
    """Tag shared by the members of one isolated group: (serial, opacity, clipped by the clip source in front of it)."""
    return (next_serial(), float(opacity), bool(clipped))   # (the process's one locked counter: unique across threads)


def _plain(leaves) -> bool:
    return all(leaf[4] == 0 and leaf[5] is None for leaf in leaves)  # (gradient entries are plain too)


def _batchable_leaves(scene: Scene, transform: Transform, linear_rgb: bool, opacity: float | None = None):
    """Memo in front of `_batchable_leaves_`: during one top-level render the same (node, transform) is asked about once per
    enclosing group that turned out not to be batchable."""
    # (only GROUP nodes are remembered: every repeated question passes through one -- a leaf, a transform or a clip over leaves is
    #  answered again faster than its key is made; material-design's 935 clip nodes went through this wrapper 29 000 times)
    memo = STATE.leaf_memo
    if memo is None or scene[0] != RENDER_GROUP:
        return _batchable_leaves_(scene, transform, linear_rgb, opacity)
    key = (id(scene), transform.key(), linear_rgb, opacity, "sub")
    if key in memo:
        return memo[key]
    res = memo[key] = _batchable_leaves_(scene, transform, linear_rgb, opacity)
    return res


def _batchable_leaves_(scene: Scene, transform: Transform, linear_rgb: bool, opacity: float | None = None):
    """[(path, m6, rule, paint4, flags, group)] when `scene` is only GROUP / TRANSFORM / solid FILL / OPACITY directly over a
    leaf, a CLIP by a single path of a leaf or of a group of plain leaves, or an OPACITY over a group of plain leaves;
    None otherwise.  flags: 0 painted, 1 clip source (coverage only), 2 clipped by the clip source right in front of it;
    group: None, or the tag of the isolated group the leaf is a member of (the device composites the members into a group
    tile and clips / fades that as a whole, svgr_batch_set_groups).  Source-over is associative, so flattening nested
    plain groups keeps the per-pixel result (to double rounding)."""
    kind, args = scene
    while kind == RENDER_TRANSFORM:   # (a chain of transforms over a node: unwrapped here, not by a call per level)
        transform = transform @ args[1]
        kind, args = scene = args[0]
    if kind == RENDER_FILL:
        path, paint, rule = args
        if paint is None:
            return []
        if rule not in _RULES:
            raise ValueError(f"Invalid fill rule: {rule}")
        if not (isinstance(paint, np.ndarray) and paint.shape == (4,)):
            if not is_gradient(paint):
                return None
            leaf = _gradient_leaf(path, paint, rule, transform, linear_rgb, opacity)
            return None if leaf is None else [leaf]
        p4 = solid_paint(paint, linear_rgb)
        if opacity is not None:
            p4 = p4 * opacity  # Layer.opacity: image * opacity (S:174)
        return [_leaf(path, transform.m6(), _RULES[rule], p4)]
    if kind == RENDER_STROKE:  # a solid stroke is a solid fill of its outline (S:666-672), nonzero rule
        path, paint, width, linecap, linejoin = args
        return _batchable_leaves(Scene.fill(_stroked(scene), paint, None), transform, linear_rgb, opacity)
    if kind == RENDER_OPACITY and opacity is None:
        target = args[0]
        while target[0] == RENDER_TRANSFORM:
            target = target[1][0]
        if target[0] in (RENDER_FILL, RENDER_STROKE):  # opacity over a single leaf commutes with the fill
            return _batchable_leaves(args[0], transform, linear_rgb, args[1])
        # OPACITY over a group of plain leaves (S:690-696): the group is composited on its own and faded as a whole
        members = _batchable_leaves(args[0], transform, linear_rgb) if _BATCH_GROUPS else None
        if members is None or not members or not _plain(members):
            return None
        tag = _new_group(args[1], False)
        return [_leaf(m[0], m[1], m[2], m[3], 0, tag, m[6]) for m in members]
    if kind == RENDER_CLIP and opacity is None and not args[2]:
        # CLIP whose target and clip are single paths: two consecutive batch entries, the clip path as a
        # coverage-only "clip source" and the fill multiplied by it (Layer.compose([mask, image], IN), S:698-715).
        # (A group under a clip is NOT the same as clipping each child: (A over B)*c != (A*c) over (B*c).)
        tgt, src = args[0], args[1]
        if tgt[0] == RENDER_FILL and src[0] == RENDER_FILL:
            # (a solid fill clipped by one path, both right here -- material-design's 935 icons --: the two entries made in
            #  this frame, not in three more; what they are is what the general route below makes)
            path, paint, rule = tgt[1]
            spath, _spaint, srule = src[1]
            if type(paint) is np.ndarray and paint.shape == (4,) and rule in _RULES and srule in _RULES:
                m6 = transform.m6()
                return [(spath, m6, _RULES[srule], _ZERO4, 1, None, None), (path, m6, _RULES[rule], solid_paint(paint, linear_rgb), 2, None, None)]
        target = _batchable_leaves_(tgt, transform, linear_rgb) if tgt[0] == RENDER_FILL else _batchable_leaves(tgt, transform, linear_rgb)
        clip_leaf = _single_mask_leaf(src, transform)
        if target is None or clip_leaf is None or not target:
            return None
        if len(target) == 1:
            t = target[0]
            if t[4] != 0 or t[5] is not None:
                return None
            return [clip_leaf, (t[0], t[1], t[2], t[3], 2, None, t[6])]
        if not _plain(target):
            return None
        if not _BATCH_GROUPS:
            return None
        # a GROUP under the clip: composited on its own, then multiplied by the clip's coverage as a whole
        tag = _new_group(1.0, True)
        return [clip_leaf] + [_leaf(t[0], t[1], t[2], t[3], 0, tag, t[6]) for t in target]
    if kind == RENDER_GROUP and opacity is None:
        out = []
        for child in args:
            # (only GROUP nodes have a memo in front of the analysis)
            sub = _batchable_leaves(child, transform, linear_rgb) if child[0] == RENDER_GROUP else _batchable_leaves_(child, transform, linear_rgb)
            if sub is None:
                return None
            out.extend(sub)
        return out
    return None


def _single_mask_leaf(scene: Scene, transform: Transform):
    """(path, m6, rule, zeros, 1) when `scene` rendered mask_only is ONE Path.mask (a FILL under transforms)."""
    kind, args = scene
    while kind == RENDER_TRANSFORM:
        transform = transform @ args[1]
        kind, args = args[0]
    if kind != RENDER_FILL:
        return None
    path, _paint, rule = args
    if rule not in _RULES:
        raise ValueError(f"Invalid fill rule: {rule}")
    return (path, transform.m6(), _RULES[rule], _ZERO4, 1, None, None)


def _effective_boxes(leaves, bboxes):
    """`effective_bboxes` as arrays: (painted leaf indices, their boxes [r0, c0, r1, c1], which of them draw something).  One
    pass over the leaves for their flags, the rest in numpy (a document's 2 000 leaves were 3 ms of tuples)."""
    n = len(leaves)
    flags = np.fromiter((leaf[4] for leaf in leaves), dtype=np.int64, count=n)
    clipped = np.fromiter((leaf[4] == 2 or (leaf[5] is not None and leaf[5][2]) for leaf in leaves), dtype=bool, count=n)
    return _effective_boxes_arrays(flags, clipped, bboxes)


def _effective_boxes_arrays(flags, clipped, bboxes):
    """`_effective_boxes` from the leaves' flags (0 painted, 1 clip source, 2 clipped) and "clipped by the source in front of it"
    (a clipped fill, a member of a clipped group) as arrays: what a display list keeps of its leaves."""
    n = len(flags)
    bb = np.asarray(bboxes, dtype=np.int64).reshape(n, 4)
    box = np.stack([bb[:, 0], bb[:, 1], bb[:, 0] + bb[:, 2], bb[:, 1] + bb[:, 3]], axis=1)
    ok = (bb[:, 2] > 0) & (bb[:, 3] > 0)
    is_src = flags == 1
    # the last clip source at or in front of every leaf (-1: none seen)
    last_src = np.maximum.accumulate(np.where(is_src, np.arange(n), -1))
    has_src = last_src >= 0
    src = np.where(has_src, last_src, 0)
    src_ok = has_src & ok[src]
    cb = box[src]
    inter = np.stack([np.maximum(box[:, 0], cb[:, 0]), np.maximum(box[:, 1], cb[:, 1]),
                      np.minimum(box[:, 2], cb[:, 2]), np.minimum(box[:, 3], cb[:, 3])], axis=1)
    inter_ok = src_ok & (inter[:, 0] < inter[:, 2]) & (inter[:, 1] < inter[:, 3])
    out_box = np.where(clipped[:, None], inter, box)
    out_ok = ok & np.where(clipped, inter_ok, True)
    painted = np.nonzero(~is_src)[0]
    return painted, out_box[painted], out_ok[painted]


def effective_bboxes(leaves, bboxes):
    """Per painted leaf the bbox its layer would have in the reference: its own clipped bbox, or for a clipped fill (and for
    the members of a clipped group) the intersection with the clip's bbox (canvas_merge_intersect, S:392-404).
    None = nothing to draw."""
    if not len(leaves):
        return []
    _painted, box, ok = _effective_boxes(leaves, bboxes)
    return [(int(b[0]), int(b[1]), int(b[2] - b[0]), int(b[3] - b[1])) if k else None for b, k in zip(box, ok)]


_PACKED_RUNS: dict = {}   # tuple of path ids -> (the paths, segs, kinds, offs): the packed segments of a run of paths, concatenated


def _packed_paths(paths):
    """(segs (S, 8), kinds (S,), offs (n + 1,)) of a list of paths: `Path.packed()` of each, concatenated.  Kept per list of path
    OBJECTS (a `Path` is a value; it has kept its own packed segments since round 1): a document's runs are the same lists of paths
    in every render, and concatenating 1 600 small arrays was a tenth of a default render of icons.svg."""
    n = len(paths)
    if n >= 8:
        key = tuple(map(id, paths))
        hit = _PACKED_RUNS.get(key)
        if hit is not None and all(a is b for a, b in zip(hit[0], paths)):
            return hit[1], hit[2], hit[3]
    # (comprehensions, not one loop with a dozen appends per leaf: material-design's 1 924 leaves were 3 ms of it)
    packs = [p.packed() for p in paths]
    seg_list = [pk[0] for pk in packs]
    offs = np.zeros(n + 1, dtype=np.int64)
    if n:
        np.cumsum(np.fromiter(map(len, seg_list), dtype=np.int64, count=n), out=offs[1:])
    segs = np.concatenate(seg_list) if n else np.zeros((0, 8))
    kinds = np.concatenate([pk[1] for pk in packs]) if n else np.zeros(0, dtype=np.uint8)
    if n >= 8:
        for a in (segs, kinds, offs):
            a.flags.writeable = False
        if len(_PACKED_RUNS) >= 64:
            _PACKED_RUNS.pop(next(iter(_PACKED_RUNS)))
        _PACKED_RUNS[key] = (list(paths), segs, kinds, offs)
    return segs, kinds, offs


def build_batch(leaves, viewport, ctx=None, row_shift=None) -> "_abi.Batch":
    """Pack paint-ordered leaves [(path, m6, rule, paint4, flags[, group[, grad]])] into one device batch.  `row_shift`: per leaf
    the rows its geometry (and its gradient's frame) is moved down by (`_shift_leaf`, for all leaves at once)."""
    ctx = ctx or _abi.Context.get()
    n = len(leaves)
    segs, kinds, offs = _packed_paths([leaf[0] for leaf in leaves])
    m6s = np.concatenate([leaf[1] for leaf in leaves]).astype(np.float64, copy=False).reshape(n, 6) if n else np.zeros((0, 6))
    if row_shift is not None and n:
        m6s[:, 2] += np.asarray(row_shift, dtype=np.float64)
    rules = np.fromiter((leaf[2] | (leaf[4] << 1) for leaf in leaves), dtype=np.uint8, count=n)  # SVGR_PATH_CLIP_SOURCE = 2, SVGR_PATH_CLIPPED = 4
    paints = np.concatenate([leaf[3] for leaf in leaves]).astype(np.float64, copy=False).reshape(n, 4) if n else np.zeros((0, 4))
    path_group, group_src, group_op, serial_to_gid = None, [], [], {}
    path_grad, grads, keep_alive, pending = None, [], [], []
    if any(len(leaf) > 5 and leaf[5] is not None for leaf in leaves):
        path_group = []
        for i, leaf in enumerate(leaves):
            group = leaf[5] if len(leaf) > 5 else None
            if group is None:
                path_group.append(-1)
                continue
            gid = serial_to_gid.get(group[0])
            if gid is None:
                gid = serial_to_gid[group[0]] = len(group_src)
                group_src.append(i - 1 if group[2] else -1)  # the clip source sits right in front of the first member
                group_op.append(group[1])
            path_group.append(gid)
    if any(len(leaf) > 6 and leaf[6] is not None for leaf in leaves):
        path_grad = []
        for i, leaf in enumerate(leaves):
            grad = leaf[6] if len(leaf) > 6 else None
            if grad is None:
                path_grad.append(-1)
            else:
                shift = row_shift[i] if row_shift is not None else 0
                if grad[0] is None:
                    # (an objectBoundingBox frame: `_resolve_frames` fills it in behind the plan; until then a valid stand-in)
                    _g0, _k0, paint, transform, lin = grad
                    pkey = (id(paint), transform.key(), lin, "stand-in")
                    hit = _GRAD_ABI_MEMO.get(pkey)
                    if hit is None or hit[0] is not paint:
                        g, keep = paint.abi(transform.invert, lin)
                        _GRAD_ABI_MEMO[pkey] = hit = (paint, g, keep)
                    pending.append((len(grads), i, paint, transform, lin, shift))
                    grad = (hit[1], hit[2])
                elif shift:
                    grad = _shift_leaf(leaf, shift)[6]   # (the gradient's frame moves with the path)
                path_grad.append(len(grads))
                grads.append(grad[0])
                keep_alive.append(grad[1])
    vp = None if viewport is None else [int(v) for v in viewport]
    batch = _abi.Batch(ctx, segs, kinds, offs, m6s, rules, paints, viewport=vp, flatness=FLATNESS)
    if group_src:
        batch.set_groups(path_group, group_src, group_op)
    if grads:
        batch.set_gradients(path_grad, grads)  # (copies the descriptions to the device before it returns)
        if pending:
            batch._frames = (path_grad, grads, pending)
    del keep_alive
    return batch


def _render_run(leaves, viewport, linear_rgb):
    """One batch -> one Layer covering the union of the leaves' bboxes (what Layer.compose of
    the individual fill layers returns, S:366-379)."""
    ctx = _abi.Context.get()
    run_plans, retain, serial = STATE.run_plans, STATE.retain, STATE.serial
    rkey = _run_key(leaves, viewport) if leaves and viewport is not None and (run_plans is not None) else None
    pre = None
    if rkey is not None:
        pre = run_plans.get(rkey) if retain is not None else run_plans.pop(rkey, None)
    if pre is not None:
        leaves, batch = pre[0], pre[1]  # (built and planned by the pre-pass of Scene.render -- of this render or of an earlier one)
        if len(pre) > 4:
            pre[4] = serial             # (an on-demand entry that came back: it stays for another render)
    else:
        leaves = _drop_empty(leaves)
        if not leaves:
            return None
        batch = build_batch(leaves, viewport, ctx)
        batch.plan()
        _resolve_frames(batch)
        if retain is not None and rkey is not None:
            # (built on demand: kept as long as its key keeps coming back, _Retained.sweep_on_demand)
            pre = run_plans[rkey] = [leaves, batch, None, None, serial]
    if pre is not None and len(pre) > 2 and pre[3] is not None:
        win, in_hull = pre[2], pre[3]   # (its window and hull membership were worked out by the pre-pass or by the first render)
    else:
        win, in_hull = _run_window(leaves, batch)
        if pre is not None and retain is not None:
            if len(pre) == 2:
                pre = run_plans[rkey] = [pre[0], pre[1], win, in_hull]
            else:
                pre[2], pre[3] = win, in_hull
    if win is None:
        if pre is None or retain is None:
            batch.destroy()
        return None
    ur0, uc0, urows, ucols = win
    # only the union of the leaves' bboxes is rendered (a render window of the batch's canvas): the tiles outside it are
    # not touched, and the layer needs no crop
    shape = (urows, ucols, 4)
    ready = getattr(batch, "ready", None)
    if ready is not None and ready[0] == serial:
        out, batch.ready = ready[1], None   # (drawn by `_prefetch_windows` together with the document's other runs)
    else:
        out = ctx.alloc(urows * ucols * 32)
        batch.render(out, _abi.OUT_CANVAS_F64, window=(ur0, uc0, urows, ucols))
    layer = Layer._from_device(out, shape, (ur0, uc0), True, linear_rgb)

    def hull_points():
        edges, edge_path = batch.all_edges()
        return edges[in_hull[edge_path]]

    return layer, ConvexHull(_source=hull_points)


def _run_window(leaves, batch):
    """(window, hull membership) of a planned run: the union of its leaves' effective bboxes (None: nothing to draw)."""
    if isinstance(batch, _RunView) and batch.shared.leaves is not None:   # (worked out for all runs of the shared batch together)
        wins, hull_all = batch.shared.run_windows()
        return wins[batch.lo], hull_all[batch.lo:batch.hi].copy()
    in_hull = np.zeros(len(leaves), dtype=bool)
    if not len(leaves):
        return None, in_hull
    if len(leaves) == 1 and leaves[0][4] == 0 and leaves[0][5] is None:   # (one plain leaf: its own bbox)
        r0, c0, rows, cols = (int(v) for v in batch.bboxes()[0])
        if rows <= 0 or cols <= 0:
            return None, in_hull
        in_hull[0] = True
        return (r0, c0, rows, cols), in_hull
    painted, box, ok = _effective_boxes(leaves, batch.bboxes())
    win = None
    if ok.any():
        v = box[ok]
        ur0, uc0 = int(v[:, 0].min()), int(v[:, 1].min())
        win = (ur0, uc0, int(v[:, 2].max()) - ur0, int(v[:, 3].max()) - uc0)
    # The group's hull merges the hulls of the children that drew something (S:676-684): a leaf whose clipped bbox is
    # empty returned None there and does not count; one that is partly visible counts with ALL its lines (S:993).  Clip
    # paths do not belong to it (S:715 returns the target's hull).
    in_hull[painted] = ok
    return win, in_hull


# (round 4 drew the windows a launch each on eight streams: 1.59 -> 0.73 ms of GPU time for icons.svg's 30, off by default because the
#  host was the bound and the up-front launches cost it 0.1-0.2 ms.  Round 5: ONE launch for up to 64 windows -- a window table in
#  the kernel argument --, on by default; SVGR_NO_WINDOW_PREFETCH draws a run's window when the walk meets it)
_PREFETCH_WINDOWS = __import__("os").environ.get("SVGR_NO_WINDOW_PREFETCH") is None
_PREFETCH_MAX_BYTES = 4 << 30   # of run layers drawn ahead of the walk; beyond it the runs are drawn when the walk meets them


def _prefetch_windows(run_plans):
    """Draw the windows of all the runs that share a batch NOW, in one launch (svgr_batch_render_windows): a run's window is a
    few dozen workgroups that live as long as its heaviest tile -- icons.svg: 30 launches, 1.6 ms one after the other, the
    longest alone 0.23 --, and no run depends on anything the walk produces.  `_render_run` hands the layers out."""
    if not _PREFETCH_WINDOWS or not run_plans:
        return
    ctx = _abi.Context.get()
    by_batch: dict = {}
    total = 0
    for entry in run_plans.values():
        view = entry[1]
        if not isinstance(view, _RunView) or view.dead:
            continue
        if len(entry) == 2:
            entry.extend(_run_window(entry[0], view))
        win = entry[2]
        if win is None:
            continue
        total += int(win[2]) * int(win[3]) * 32
        by_batch.setdefault(id(view.shared), (view.shared, []))[1].append((view, win))
    if total > _PREFETCH_MAX_BYTES:
        return
    serial = STATE.serial
    for shared, items in by_batch.values():
        if len(items) < 2:
            continue
        outs = [ctx.alloc(int(w[2]) * int(w[3]) * 32) for _v, w in items]
        wins = [(w[0] + v.shift, w[1], w[2], w[3]) for v, w in items]
        shared.batch.render_windows(outs, _abi.OUT_CANVAS_F64, wins, _abi.RENDER_SAME_GEOMETRY if shared.serial == serial else 0)
        shared.serial = serial
        for (v, _w), out in zip(items, outs):
            v.ready = (serial, out)


def _drop_empty(leaves):
    """Remove leaves without segments.  A clip source goes together with what it clips (the clipped fill, or the members of
    the clipped group behind it): without the source nothing of them is visible, without them the source is not needed."""
    has_segs = lambda leaf: len(leaf[0].packed()[0]) > 0  # noqa: E731
    if all(len(leaf[0].packed()[0]) for leaf in leaves):   # (nothing to drop: the usual case)
        return list(leaves)
    out, i = [], 0
    while i < len(leaves):
        leaf = leaves[i]
        if leaf[4] == 1:  # clip source + its dependants
            j = i + 1
            if j < len(leaves) and leaves[j][4] == 2:
                j += 1
            else:
                tag = leaves[j][5] if j < len(leaves) else None
                while j < len(leaves) and leaves[j][5] is not None and leaves[j][5] is tag and tag[2]:
                    j += 1
            deps = [l for l in leaves[i + 1:j] if has_segs(l)]
            if has_segs(leaf) and deps:
                out.append(leaf)
                out.extend(deps)
            i = j
            continue
        if has_segs(leaf):
            out.append(leaf)
        i += 1
    return out


def render_canvas(scene_or_leaves, transform: Transform | None, viewport, linear_rgb: bool = False,
                  out_f64: bool = False, clip01: bool = True):
    """Production entry: whole scene -> (rows, cols, 4) premultiplied canvas over `viewport`,
    float32 by default.  Equivalent to the reference CLI's render + ``canvas_merge_at`` on a zero
    canvas (S:3857-3875) for scenes made of solid fills; returns (canvas ndarray, stats)."""
    if isinstance(scene_or_leaves, Scene):
        leaves = _batchable_leaves(scene_or_leaves, transform, linear_rgb)
        if leaves is None:
            raise NotImplementedError("scene contains nodes outside the batched path; use Scene.render")
    else:
        leaves = scene_or_leaves
    ctx = _abi.Context.get()
    batch = build_batch(leaves, viewport, ctx)
    st = batch.plan()
    _resolve_frames(batch)
    rows, cols = int(viewport[2]), int(viewport[3])
    out = ctx.alloc(rows * cols * (32 if out_f64 else 16))
    batch.render(out, _abi.OUT_CANVAS_F64 if out_f64 else _abi.OUT_CANVAS_F32, _abi.RENDER_CLIP01 if clip01 else 0)
    img = out.download((rows, cols, 4), np.float64 if out_f64 else np.float32)
    batch.destroy()
    return img, st
