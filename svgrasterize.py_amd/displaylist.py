"""A scene that is batch entries and nothing else, compiled ONCE to flat arrays ("Scene -> (once, Python) arrays").

`Scene.render` (S:649-752) walks the tree on every call; what the walk finds out about a document made of solid fills, strokes,
transforms, single-path clips and opacity / clip groups -- which leaves there are, in which order, with which rule, flags and
isolated group -- does not depend on the render transform at all.  Only the leaves' matrices do: `transform @ A @ B ...` down the
chain of TRANSFORM nodes above each leaf.  This module keeps the transform-independent part per scene object (the analysis of
`scene._batchable_leaves_`, leaf for leaf, and `build_batch`'s concatenations) and makes a render of such a scene

    chain products (stacked 3x3 matmuls, level by level: the same BLAS call on the same operands in the same association as the
    walk's `Transform.__matmul__` -- bit-identical, checked in tests/test_displaylist.py) -> `_abi.Batch` -> plan -> window -> Layer

instead of ~2 000 Python frames of tree walk and ~10 000 small-array operations (material-design @4096: 4.7 -> ~1 ms per render).
Scenes with anything else in them (gradients, filters, masks, patterns, objectBoundingBox clips) are not display lists: `compile`
returns None and `Scene.render` takes its general route.

What is kept is keyed by the scene OBJECT (a `Scene` is an immutable tuple tree; the entry holds it, so its id cannot come back as
another scene's) and is transform-, viewport- and device-independent; nothing of a render's RESULT is kept.  Paint arrays are read
again at every render (a colour edited in place is drawn with its new value, as the reference would); path geometry is a value
type here as in the reference (`Path.packed` has kept its packed segments since round 1).
"""
from __future__ import annotations

import os
import threading

import numpy as np

from . import _abi
from .geometry import ConvexHull, FLATNESS, Transform, _RULES, solid_paint
from .layer import Layer

ENABLED = os.environ.get("SVGR_NO_DISPLAY_LISTS") is None
_MAX_ENTRIES = 32
_CACHE: "dict[int, tuple]" = {}     # id(scene) -> (scene, {linear_rgb: DisplayList or None})   (insertion-ordered: oldest first)
_LOCK = threading.Lock()
_ZERO4 = np.zeros(4)


class _NotFlat(Exception):
    """The scene is not batch entries only."""


class DisplayList:
    __slots__ = ("n", "paths", "segs", "kinds", "offs", "rules", "flags", "clipped", "paint_refs", "paint_uniq", "opac", "raw", "paints",
                 "linear_rgb", "path_group", "group_src", "group_op", "leaf_node", "levels", "n_nodes")

    # -- per render ---------------------------------------------------------------------------------------------------------------
    def matrices(self, transform: Transform) -> np.ndarray:
        """(n, 6): every leaf's accumulated matrix `transform @ A @ B ...`, the products made level by level over all chains at once
        (np.matmul on stacked 3x3 matrices calls the BLAS routine `np.dot` calls, per pair: the walk's own bits)."""
        P = np.empty((self.n_nodes, 3, 3), dtype=np.float64)
        P[0] = np.asarray(transform.m, dtype=np.float64)
        for nodes, parents, mats in self.levels:
            P[nodes] = np.matmul(P[parents], mats)
        return np.ascontiguousarray(P[self.leaf_node, :2, :]).reshape(self.n, 6)

    def current_paints(self) -> np.ndarray:
        """(n, 4) paints in the compositing space; converted again only when some paint array changed since the last render."""
        # (the distinct paint OBJECTS are read -- a document's clip sources all share one --, not one array per leaf)
        raw = np.concatenate(self.paint_uniq) if self.paint_uniq else np.zeros(0)
        if self.paints is None or not np.array_equal(raw, self.raw):
            out = np.empty((self.n, 4), dtype=np.float64)
            lin = self.linear_rgb
            for i, (p, o) in enumerate(zip(self.paint_refs, self.opac)):
                if p is _ZERO4:
                    out[i] = 0.0
                else:
                    c = solid_paint(p, lin)
                    out[i] = c if o is None else c * o   # Layer.opacity over the leaf: image * opacity (S:174)
            self.raw, self.paints = raw, out
        return self.paints

    def render(self, transform: Transform, viewport, linear_rgb: bool):
        """What `Scene._render` returns for this scene: (Layer over the union of the leaves' effective bboxes, lazy hull) or None."""
        from .scene import _effective_boxes_arrays  # noqa: PLC0415

        if self.n == 0:
            return None
        ctx = _abi.Context.get()
        vp = [int(v) for v in viewport]
        batch = _abi.Batch(ctx, self.segs, self.kinds, self.offs, self.matrices(transform), self.rules, self.current_paints(),
                           viewport=vp, flatness=FLATNESS)
        try:
            if self.group_src is not None:
                batch.set_groups(self.path_group, self.group_src, self.group_op)
            batch.plan()
            painted, box, ok = _effective_boxes_arrays(self.flags, self.clipped, batch.bboxes())
        except Exception:
            batch.destroy()
            raise
        if not ok.any():
            batch.destroy()
            return None
        v = box[ok]
        ur0, uc0 = int(v[:, 0].min()), int(v[:, 1].min())
        urows, ucols = int(v[:, 2].max()) - ur0, int(v[:, 3].max()) - uc0
        in_hull = np.zeros(self.n, dtype=bool)
        in_hull[painted] = ok
        out = ctx.alloc(urows * ucols * 32)
        batch.render(out, _abi.OUT_CANVAS_F64, window=(ur0, uc0, urows, ucols))
        layer = Layer._from_device(out, (urows, ucols, 4), (ur0, uc0), True, linear_rgb)

        def hull_points():
            edges, edge_path = batch.all_edges()
            return edges[in_hull[edge_path]]

        return layer, ConvexHull(_source=hull_points)


def _compile(scene, linear_rgb: bool):
    """The analysis of `scene._batchable_leaves_` with the transform left symbolic (a chain of TRANSFORM matrices per leaf), for
    solid paints only, then `scene._drop_empty` and `scene.build_batch`'s packing.  None: not a display list."""
    from . import scene as sc  # noqa: PLC0415

    groups: list = []   # (opacity, clipped) per isolated group

    def leaf(path, chain, rule, paint, opacity, flags=0, group=-1):
        return [path, chain, _RULES[rule], paint, opacity, flags, group]

    def plain(leaves):
        return all(lf[5] == 0 and lf[6] < 0 for lf in leaves)

    def walk(node, chain, opacity=None):
        kind, args = node
        while kind == sc.RENDER_TRANSFORM:
            chain = chain + (args[1],)
            kind, args = node = args[0]
        if kind == sc.RENDER_FILL:
            path, paint, rule = args
            if paint is None:
                return []
            if rule not in _RULES:
                raise ValueError(f"Invalid fill rule: {rule}")
            if not (isinstance(paint, np.ndarray) and paint.shape == (4,)):
                raise _NotFlat
            return [leaf(path, chain, rule, paint, opacity)]
        if kind == sc.RENDER_STROKE:
            path, paint, _w, _cap, _join = args
            return walk(sc.Scene.fill(sc._stroked(node), paint, None), chain, opacity)
        if kind == sc.RENDER_OPACITY and opacity is None:
            target = args[0]
            while target[0] == sc.RENDER_TRANSFORM:
                target = target[1][0]
            if target[0] in (sc.RENDER_FILL, sc.RENDER_STROKE):
                return walk(args[0], chain, args[1])
            if not sc._BATCH_GROUPS:
                raise _NotFlat
            members = walk(args[0], chain)
            if not members or not plain(members):
                raise _NotFlat
            groups.append((float(args[1]), False))
            gid = len(groups) - 1
            for m in members:
                m[6] = gid
            return members
        if kind == sc.RENDER_CLIP and opacity is None and not args[2]:
            tgt, src = args[0], args[1]
            target = walk(tgt, chain)
            # the clip as ONE Path.mask: a FILL under transforms (scene._single_mask_leaf)
            skind, sargs, schain = src[0], src[1], chain
            while skind == sc.RENDER_TRANSFORM:
                schain = schain + (sargs[1],)
                skind, sargs = sargs[0]
            if skind != sc.RENDER_FILL:
                raise _NotFlat
            spath, _spaint, srule = sargs
            if srule not in _RULES:
                raise ValueError(f"Invalid fill rule: {srule}")
            if not target:
                raise _NotFlat
            clip_leaf = leaf(spath, schain, srule, _ZERO4, None, 1)
            if len(target) == 1:
                t = target[0]
                if t[5] != 0 or t[6] >= 0:
                    raise _NotFlat
                t[5] = 2
                return [clip_leaf, t]
            if not plain(target) or not sc._BATCH_GROUPS:
                raise _NotFlat
            groups.append((1.0, True))
            gid = len(groups) - 1
            for t in target:
                t[6] = gid
            return [clip_leaf] + target
        if kind == sc.RENDER_GROUP and opacity is None:
            out = []
            for child in args:
                out.extend(walk(child, chain))
            return out
        raise _NotFlat

    try:
        leaves = walk(scene, ())
    except _NotFlat:
        return None
    # scene._drop_empty: leaves without segments go; a clip source goes together with what it clips
    has = [len(lf[0].packed()[0]) > 0 for lf in leaves]
    if not all(has):
        kept, i = [], 0
        while i < len(leaves):
            lf = leaves[i]
            if lf[5] == 1:
                j = i + 1
                if j < len(leaves) and leaves[j][5] == 2:
                    j += 1
                else:
                    g = leaves[j][6] if j < len(leaves) else -1
                    while j < len(leaves) and leaves[j][6] >= 0 and leaves[j][6] == g and groups[g][1]:
                        j += 1
                deps = [leaves[k] for k in range(i + 1, j) if has[k]]
                if has[i] and deps:
                    kept.append(lf)
                    kept.extend(deps)
                i = j
                continue
            if has[i]:
                kept.append(lf)
            i += 1
        leaves = kept
    dl = DisplayList()
    n = dl.n = len(leaves)
    dl.linear_rgb = bool(linear_rgb)
    dl.paths = [lf[0] for lf in leaves]
    packs = [p.packed() for p in dl.paths]
    dl.offs = np.zeros(n + 1, dtype=np.int64)
    if n:
        np.cumsum(np.fromiter((len(pk[0]) for pk in packs), dtype=np.int64, count=n), out=dl.offs[1:])
    dl.segs = np.ascontiguousarray(np.concatenate([pk[0] for pk in packs])) if n else np.zeros((0, 8))
    dl.kinds = np.ascontiguousarray(np.concatenate([pk[1] for pk in packs])) if n else np.zeros(0, dtype=np.uint8)
    dl.flags = np.fromiter((lf[5] for lf in leaves), dtype=np.int64, count=n)
    dl.rules = np.fromiter((lf[2] | (lf[5] << 1) for lf in leaves), dtype=np.uint8, count=n)   # SVGR_PATH_CLIP_SOURCE = 2, SVGR_PATH_CLIPPED = 4
    dl.clipped = np.fromiter((lf[5] == 2 or (lf[6] >= 0 and groups[lf[6]][1]) for lf in leaves), dtype=bool, count=n)
    dl.paint_refs = [lf[3] for lf in leaves]
    dl.paint_uniq = list({id(p): p for p in dl.paint_refs}.values())
    dl.opac = [lf[4] for lf in leaves]
    dl.raw = dl.paints = None
    # isolated groups in order of first appearance (scene.build_batch): the clip source sits right in front of the first member
    dl.path_group = dl.group_src = dl.group_op = None
    if any(lf[6] >= 0 for lf in leaves):
        gid_of, pg, gsrc, gop = {}, [], [], []
        for i, lf in enumerate(leaves):
            if lf[6] < 0:
                pg.append(-1)
                continue
            g = gid_of.get(lf[6])
            if g is None:
                g = gid_of[lf[6]] = len(gsrc)
                gsrc.append(i - 1 if groups[lf[6]][1] else -1)
                gop.append(groups[lf[6]][0])
            pg.append(g)
        dl.path_group, dl.group_src, dl.group_op = pg, gsrc, gop
    # the chains as a trie of prefixes: node 0 = the render transform, a node per distinct (prefix, next matrix object)
    node_of = {(): 0}
    by_level: list = []
    leaf_node = np.zeros(n, dtype=np.int64)
    for i, lf in enumerate(leaves):
        chain = lf[1]
        key = ()
        at = 0
        for d, tr in enumerate(chain):
            key = key + (id(tr),)
            nxt = node_of.get(key)
            if nxt is None:
                nxt = node_of[key] = len(node_of)
                while len(by_level) <= d:
                    by_level.append(([], [], []))
                by_level[d][0].append(nxt)
                by_level[d][1].append(at)
                by_level[d][2].append(np.array(tr.m, dtype=np.float64).reshape(3, 3))
            at = nxt
        leaf_node[i] = at
    dl.leaf_node = leaf_node
    dl.n_nodes = len(node_of)
    dl.levels = [(np.asarray(a, dtype=np.int64), np.asarray(b, dtype=np.int64), np.stack(c)) for a, b, c in by_level]
    return dl


def get(scene, linear_rgb: bool):
    """The display list of `scene` in this colour space, compiled on first use; None when the scene is not one."""
    if not ENABLED:
        return None
    lin = bool(linear_rgb)
    with _LOCK:
        hit = _CACHE.get(id(scene))
        if hit is not None and hit[0] is scene and lin in hit[1]:
            return hit[1][lin]
    dl = _compile(scene, lin)     # (raises what the walk would raise: an invalid fill rule)
    with _LOCK:
        hit = _CACHE.get(id(scene))
        if hit is None or hit[0] is not scene:
            hit = _CACHE[id(scene)] = (scene, {})
            while len(_CACHE) > _MAX_ENTRIES:
                _CACHE.pop(next(iter(_CACHE)))
        hit[1][lin] = dl
    return dl


def clear() -> None:
    with _LOCK:
        _CACHE.clear()
