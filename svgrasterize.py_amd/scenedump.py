"""Load the scene dumps under tests/golden/scene_*.npz (format: oracle/gen_golden.py, class Dumper).

A dump is DATA extracted from an SVG by the reference's front-end in the build container: the node
tree (JSON) plus per-leaf user-space lines / cubics, so the GPU box needs neither the SVG parser
nor the stroker (both out of scope)."""
from __future__ import annotations

import json

import numpy as np

from .geometry import Path, Transform
from .paint import GradLinear, GradRadial
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
