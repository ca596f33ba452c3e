"""Oracle tooling: load the upstream reference in THIS build container.

TEST INFRASTRUCTURE ONLY. Nothing under ``svgrasterize.py_amd/`` may import this.
It only runs where ``/root/reference`` exists (never on the GPU box); its single
purpose is to let ``oracle/gen_golden.py`` produce the data fixtures committed
under ``tests/golden/``.

The reference needs Python >= 3.12 (PEP 695 ``type X = ...`` aliases,
svgrasterize.py:55-58, 586-595, 883-892). This interpreter is 3.10, so the source
text is compiled from memory with those alias statements commented out and the
names bound to ``typing.Any``. All annotations are lazy
(``from __future__ import annotations``, svgrasterize.py:20), so behaviour is
unchanged. No reference source is copied into this repository.
"""
from __future__ import annotations

import os
import re
import sys
import types

REF_DIR = os.environ.get("SVGR_REFERENCE_DIR", "/root/reference")
REF_SRC = os.path.join(REF_DIR, "svgrasterize.py")
_MOD_NAME = "svgrasterize_ref"


def available() -> bool:
    return os.path.isfile(REF_SRC)


def load() -> types.ModuleType:
    """Return the reference module (cached in ``sys.modules``)."""
    if _MOD_NAME in sys.modules:
        return sys.modules[_MOD_NAME]
    if not available():
        raise FileNotFoundError(f"reference not present at {REF_SRC}")
    with open(REF_SRC) as fh:
        lines = fh.read().split("\n")
    out: list[str] = []
    i = 0
    alias = re.compile(r"^type (\w+) = (.*)$")
    while i < len(lines):
        m = alias.match(lines[i])
        if m is None:
            out.append(lines[i])
        else:
            if m.group(2).strip() == "(":  # parenthesised multi-line alias
                while lines[i].strip() != ")":
                    out.append("# " + lines[i])
                    i += 1
            out.append("# " + lines[i])
            out.append(f"{m.group(1)} = Any")
        i += 1
    mod = types.ModuleType(_MOD_NAME)
    mod.__file__ = REF_SRC  # DEFAULT_FONTS resolves relative to __file__
    sys.modules[_MOD_NAME] = mod
    exec(compile("\n".join(out), REF_SRC, "exec"), mod.__dict__)
    return mod


def render_like_cli(svg_path: str, width: int | None = None, linear_rgb: bool = False):
    """Render the way the reference CLI does (svgrasterize.py:3819-3875).

    Returns ``(scene, (h, w), layer, hull, canvas)`` where ``canvas`` is the final
    premultiplied float64 ``(h, w, 4)`` array after ``canvas_merge_at``.
    """
    import numpy as np

    ref = load()
    fonts = ref.FontsDB()
    fonts.register_file(ref.DEFAULT_FONTS)
    tr = ref.Transform().matrix(0, 1, 0, 1, 0, 0)
    scene, _ids, size = ref.svg_scene_from_filepath(svg_path, width=width, fonts=fonts)
    w, h = size
    result = scene.render(tr, viewport=[0, 0, int(h), int(w)], linear_rgb=linear_rgb)
    layer, hull = result
    canvas = np.zeros((int(h), int(w), 4), dtype=np.float64)
    out = layer.convert(pre_alpha=True, linear_rgb=linear_rgb)
    ref.canvas_merge_at(canvas, out.image, out.offset)
    return scene, (int(h), int(w)), layer, hull, canvas
