"""Load the scene dumps under tests/golden/scene_*.npz (format: oracle/gen_golden.py, class Dumper).

A dump is DATA extracted from an SVG by the reference's front-end in the build container: the node
tree (JSON) plus per-leaf user-space lines / cubics, so the GPU box needs neither the SVG parser
nor the stroker (both out of scope)."""
from __future__ import annotations

import json

import numpy as np

from .geometry import Path, Transform
from .paint import GradLinear, GradRadial, Pattern
from .scene import Scene


def _tr(m6):
    m = np.eye(3)
    m[:2, :] = np.array(m6, dtype=np.float64).reshape(2, 3)
    return Transform(m)


def load_scene(npz_path: str):
    """Returns (Scene, info dict, npz handle)."""
    z = np.load(npz_path, allow_pickle=False)
    tree = json.loads(str(z["tree"]))
    info = json.loads(str(z["info"]))
    lines, cubics, loff, coff = z["lines"], z["cubics"], z["line_off"], z["cubic_off"]

    def path(i):
        return Path.from_arrays(lines[loff[i]: loff[i + 1]], cubics[coff[i]: coff[i + 1]])

    def paint(p):
        if p is None:
            return None
        if p["k"] == "rgba":
            return np.array(p["v"], dtype=np.float64)
        if p["k"] == "pattern":
            return Pattern(node(p["scene"]), p["scene_bbox_units"], p["scene_view_box"], *p["cell"], _tr(p["tr"]), p["bbox_units"])
        tr = None if p.get("tr") is None else _tr(p["tr"])
        stops = [(o, np.array(c)) for o, c in p["stops"]]
        if p["k"] == "linear":
            return GradLinear(np.array(p["p0"]), np.array(p["p1"]), stops, tr, p["spread"], p["bbox_units"], p["linear_rgb"])
        if p["k"] == "radial":
            arr = lambda v: None if v is None else np.array(v)
            return GradRadial(arr(p["center"]), p["radius"], arr(p["fcenter"]), p["fradius"], stops, tr, p["spread"],
                              p["bbox_units"], p["linear_rgb"])
        raise NotImplementedError(p["k"])

    def node(n):
        t = n["t"]
        if t == "fill":
            return Scene.fill(path(n["path"]), paint(n["paint"]), n["rule"])
        if t == "group":
            return Scene(2, tuple(node(c) for c in n["c"]))
        if t == "opacity":
            return Scene(3, (node(n["c"]), n["o"]))
        if t == "clip":
            return Scene(4, (node(n["c"]), node(n["clip"]), n["bbox_units"]))
        if t == "mask":
            return Scene(5, (node(n["c"]), node(n["mask"]), n["bbox_units"]))
        if t == "transform":
            return Scene(6, (node(n["c"]), _tr(n["m"])))
        if t == "filter":
            from .filters import Filter

            flt = Filter({"SourceAlpha": 0, "SourceGraphic": 1},
                         [(f["type"], tuple(f["attrs"]) if isinstance(f["attrs"], list) else f["attrs"], list(f["inputs"]))
                          for f in n["filters"]])
            return Scene(7, (node(n["c"]), flt))
        raise ValueError(t)

    return node(tree), info, z


def gather_path(path: Path):
    """``(lines (N, 2, 2), cubics (M, 4, 2))`` of a path, in segment order: what Path.mask hands to the device."""
    segs, kinds = path.packed()
    return segs[kinds == 0][:, :4].reshape(-1, 2, 2), segs[kinds == 1].reshape(-1, 4, 2)


def dump_scene(scene: Scene):
    """The inverse of `load_scene`: a Scene as plain data in the dump format of oracle/gen_golden.py (class Dumper):
    ``(tree, arrays)`` with arrays = lines (N, 2, 2), cubics (M, 4, 2), line_off, cubic_off (per path index).
    STROKE nodes are stored as FILL nodes of their stroke outline (``from_stroke``), exactly like the reference dumps."""
    from .filters import FE_GAUSSIAN_BLUR
    from .scene import (RENDER_CLIP, RENDER_FILL, RENDER_FILTER, RENDER_GROUP, RENDER_MASK, RENDER_OPACITY, RENDER_STROKE,
                        RENDER_TRANSFORM)

    lines, cubics, loff, coff = [], [], [0], [0]

    def add_path(path: Path) -> int:
        l, c = gather_path(path)
        lines.append(l)
        cubics.append(c)
        loff.append(loff[-1] + len(l))
        coff.append(coff[-1] + len(c))
        return len(loff) - 2

    def paint(p):
        if p is None:
            return None
        if isinstance(p, np.ndarray):
            return dict(k="rgba", v=[float(x) for x in p])
        if isinstance(p, Pattern):
            return dict(k="pattern", scene=node(p.scene), scene_bbox_units=bool(p.scene_bbox_units),
                        scene_view_box=None if p.scene_view_box is None else [float(x) for x in p.scene_view_box],
                        cell=[float(p.x), float(p.y), float(p.width), float(p.height)],
                        tr=[float(x) for x in p.transform.m[:2].ravel()], bbox_units=bool(p.bbox_units))
        if not isinstance(p, (GradLinear, GradRadial)):
            return dict(k="unsupported", name=type(p).__name__)
        common = dict(stops=[[float(o), [float(x) for x in c]] for o, c in p.stops],
                      tr=None if p.transform is None else [float(x) for x in p.transform.m[:2].ravel()],
                      spread=p.spread, bbox_units=bool(p.bbox_units), linear_rgb=p.linear_rgb)
        if isinstance(p, GradLinear):
            return dict(k="linear", p0=[float(x) for x in p.p0], p1=[float(x) for x in p.p1], **common)
        if isinstance(p, GradRadial):
            vec = lambda v: None if v is None else [float(x) for x in v]
            return dict(k="radial", center=vec(p.center), radius=None if p.radius is None else float(p.radius),
                        fcenter=vec(p.fcenter), fradius=None if p.fradius is None else float(p.fradius), **common)
        return dict(k="unsupported", name=type(p).__name__)

    def node(s: Scene):
        kind, a = s
        if kind == RENDER_FILL:
            return dict(t="fill", path=add_path(a[0]), paint=paint(a[1]), rule=a[2])
        if kind == RENDER_STROKE:
            path, pnt, width, cap, join = a
            return dict(t="fill", path=add_path(path.stroke(width, cap, join)), paint=paint(pnt), rule=None, from_stroke=True)
        if kind == RENDER_GROUP:
            return dict(t="group", c=[node(ch) for ch in a])
        if kind == RENDER_OPACITY:
            return dict(t="opacity", c=node(a[0]), o=float(a[1]))
        if kind == RENDER_CLIP:
            return dict(t="clip", c=node(a[0]), clip=node(a[1]), bbox_units=bool(a[2]))
        if kind == RENDER_MASK:
            return dict(t="mask", c=node(a[0]), mask=node(a[1]), bbox_units=bool(a[2]))
        if kind == RENDER_TRANSFORM:
            return dict(t="transform", c=node(a[0]), m=[float(x) for x in a[1].m[:2].ravel()])
        if kind == RENDER_FILTER:
            fl = []
            for ftype, attrs, inputs in a[1].filters:
                fl.append(dict(type=int(ftype), attrs=[None if v is None else float(v) for v in attrs]
                               if ftype == FE_GAUSSIAN_BLUR else repr(attrs), inputs=[int(i) for i in inputs]))
            return dict(t="filter", c=node(a[0]), filters=fl)
        raise ValueError(kind)

    tree = node(scene)
    cat = lambda xs, shape: np.concatenate(xs) if xs else np.zeros(shape)
    arrays = dict(lines=cat(lines, (0, 2, 2)), cubics=cat(cubics, (0, 4, 2)), line_off=np.array(loff, dtype=np.int64),
                  cubic_off=np.array(coff, dtype=np.int64))
    return tree, arrays


def compare_dumps(tree_a, arrays_a, tree_b, arrays_b, tol: float = 1e-12) -> list:
    """Differences between two scene dumps as a list of strings (empty: the scenes are the same up to `tol`)."""
    diffs: list = []

    def geometry(arrays, i):
        return (arrays["lines"][arrays["line_off"][i]: arrays["line_off"][i + 1]],
                arrays["cubics"][arrays["cubic_off"][i]: arrays["cubic_off"][i + 1]])

    def close(x, y):
        if x is None or y is None:
            return x is None and y is None
        x, y = np.asarray(x, dtype=np.float64), np.asarray(y, dtype=np.float64)
        return x.shape == y.shape and bool(np.all(np.abs(x - y) <= tol * (1 + np.abs(y))))

    def walk(a, b, where):
        if len(diffs) > 20:
            return
        if type(a) is not type(b):
            diffs.append(f"{where}: {type(a).__name__} vs {type(b).__name__}")
        elif isinstance(a, dict):
            if set(a) != set(b):
                diffs.append(f"{where}: keys {sorted(a)} vs {sorted(b)}")
                return
            for k in a:
                if k == "path" and a.get("t") == "fill":
                    (la, ca), (lb, cb) = geometry(arrays_a, a[k]), geometry(arrays_b, b[k])
                    if not (close(la, lb) and close(ca, cb)):
                        diffs.append(f"{where}.path: geometry differs ({la.shape}/{ca.shape} vs {lb.shape}/{cb.shape})")
                else:
                    walk(a[k], b[k], f"{where}.{k}")
        elif isinstance(a, list):
            if len(a) != len(b):
                diffs.append(f"{where}: {len(a)} vs {len(b)} items")
                return
            if a and all(isinstance(v, (int, float)) for v in a + b):
                if not close(a, b):
                    diffs.append(f"{where}: {a} vs {b}")
                return
            for i, (x, y) in enumerate(zip(a, b)):
                walk(x, y, f"{where}[{i}]")
        elif isinstance(a, float):
            if not close(a, b):
                diffs.append(f"{where}: {a} vs {b}")
        elif a != b:
            diffs.append(f"{where}: {a!r} vs {b!r}")

    walk(tree_a, tree_b, "scene")
    return diffs
