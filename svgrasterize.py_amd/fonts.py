"""SVG fonts: ``<font>`` glyph tables -> text outlines (reference ``Glyph`` / ``Font`` / ``FontsDB``, S:2563-2718).

Host code in front of the hot path: a string becomes one ``Path`` of glyph outlines, which is then filled / stroked like
any other path.  Only SVG fonts are understood (``<font>`` elements inside a document, or whole documents of them
registered with ``FontsDB.register_file`` and loaded on first use) -- the same restriction the reference has.
"""
from __future__ import annotations

import os
import warnings

from .geometry import PATH_ARC, Path

FONT_STYLE_NORMAL = "normal"
# generic families the well-known names fall back to (S:2653-2655)
_GENERIC = (
    ("sans", {"arial", "verdana"}),
    ("serif", {"times new roman", "times", "georgia"}),
    ("mono", {"iosevka", "courier", "pragmatapro"}),
)


class Glyph:
    """One glyph: its outline is kept as path data and parsed on first use."""

    __slots__ = ["unicode", "advance", "name", "path_source", "_path"]

    def __init__(self, unicode, advance: float, path_source: str, name=None):
        self.unicode = unicode
        self.advance = advance
        self.name = name
        self.path_source = path_source
        self._path = None

    @property
    def path(self) -> Path:
        if self._path is None:
            self._path = Path.from_svg(self.path_source)
        return self._path

    def __repr__(self) -> str:
        return f"Glyph(unicode={self.unicode}, name={self.name})"


class Font:
    """Glyph table of one face.  Glyph space is y-up with ``units_per_em`` units per em."""

    __slots__ = ["family", "weight", "style", "ascent", "descent", "units_per_em", "glyphs", "missing_glyph", "hkern"]

    def __init__(self, family, weight, style, ascent, descent, units_per_em, glyphs=None, missing_glyph=None, hkern=None):
        self.family, self.weight, self.style = family, weight, style
        self.ascent, self.descent, self.units_per_em = ascent, descent, units_per_em
        self.glyphs = {} if glyphs is None else glyphs
        self.missing_glyph = missing_glyph
        self.hkern = {} if hkern is None else hkern

    def str_to_glyphs(self, string: str):
        """``([(pen x, glyph)], total advance)`` in glyph units (S:2602-2634).

        Ligatures: the key is grown one character at a time while it keeps naming a glyph; a single unknown character
        maps to the missing glyph.  Kerning is subtracted from the pen before the right glyph of a pair is placed.
        """
        placed, pen, prev = [], 0.0, None
        i, n = 0, len(string)
        while i < n:
            j = i + 1
            glyph = self.glyphs.get(string[i:j])
            if glyph is None:
                glyph = self.missing_glyph
            else:
                while j < n:
                    longer = self.glyphs.get(string[i:j + 1])
                    if longer is None:
                        break
                    glyph, j = longer, j + 1
            assert glyph is not None, "font has no missing-glyph"
            i = j
            if prev is not None:
                kern = self.hkern.get((prev, glyph.unicode))
                if kern is not None:
                    pen -= kern
            placed.append((pen, glyph))
            pen += glyph.advance
            prev = glyph.unicode
        return placed, pen

    def str_to_path(self, size: float, string: str):
        """Outline of ``string`` at ``size`` user units per em, y flipped to the SVG's y-down: ``(Path, advance)``
        (S:2636-2650; ``(x + pen) * scale``, ``-y * scale`` in that order of operations)."""
        scale = size / self.units_per_em
        placed, advance = self.str_to_glyphs(string)
        subpaths = []
        for pen, glyph in placed:
            for outline in glyph.path:
                sub = []
                for kind, pts in outline:
                    assert kind != PATH_ARC
                    sub.append((kind, [[(x + pen) * scale, -y * scale] for x, y in pts]))
                subpaths.append(sub)
        return Path(subpaths), advance * scale

    def names(self) -> dict:
        return {g.name: g.unicode for g in self.glyphs.values()}

    def __repr__(self) -> str:
        return f'Font(family="{self.family}", weight={self.weight}, style={self.style}, glyphs_count={len(self.glyphs)})'


class FontsDB:
    """Fonts by lower-cased family name, with the reference's resolution order (S:2661-2718)."""

    __slots__ = ["fonts", "fonts_files"]

    def __init__(self):
        self.fonts: dict = {}
        self.fonts_files: list = []

    def register(self, font: Font, alias=None) -> None:
        self.fonts.setdefault(font.family.lower(), []).append(font)
        if alias is not None and alias != font.family:
            self.fonts.setdefault(alias.lower(), []).append(font)

    def register_file(self, path: str) -> None:
        """Remember an SVG document of ``<font>`` elements; it is loaded by the first ``resolve``."""
        self.fonts_files.append(path)

    def _load_pending(self) -> None:
        from .svg import svg_scene_from_filepath  # the loader registers every <font> it meets
        while self.fonts_files:
            source = self.fonts_files.pop()
            if not os.path.isfile(source):
                warnings.warn(f"failed to find fonts file: {source}")
                continue
            svg_scene_from_filepath(source, fonts=self)

    def resolve(self, family, weight=None, style=None):
        """Best face for ``family`` or None: exact family, else its generic family (unknown names count as serif);
        then the requested style (else normal); then the nearest weight (first registered wins ties)."""
        self._load_pending()
        family = "serif" if family is None else family.lower()
        faces = self.fonts.get(family)
        if faces is None:
            generic = "serif"
            for key, members in _GENERIC:
                if key in family or family in members:
                    generic = "monospace" if key == "mono" else key
                    break
            # the second lookup name is the reference's (S:2699); it only matters when the generic family is absent
            faces = self.fonts.get(generic, self.fonts.get("seif"))
        if faces is None:
            return None
        style = style or FONT_STYLE_NORMAL
        styled = [f for f in faces if f.style == style] or [f for f in faces if f.style == FONT_STYLE_NORMAL]
        if not styled:
            return None
        weight = weight or 400
        return min(styled, key=lambda f: abs(f.weight - weight))
