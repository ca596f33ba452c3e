"""SVG front-end: an SVG document -> ``Scene`` (reference ``svg_scene``, S:2803-3625).

Host code: XML in, scene tree out -- the part of the reference that sits in front of the hot path.  It produces
the same tree the reference's loader produces (same node nesting, same paints, same shape-to-path conversions,
the same quirks: gradients do not follow ``href``, shapes go through ``%g``-formatted path data, ``opacity`` makes an
isolated group, the order filter -> opacity -> clip-path -> mask -> transform), so that a scene loaded here renders
exactly like the scene dumps extracted from the reference (``tests/test_svg_loader.py`` compares the trees).

Supported: svg (nested, viewBox), g, defs, path, rect, circle, ellipse, line, polyline, polygon, use,
linearGradient / radialGradient / stop, pattern, clipPath, mask, filter (feOffset, feGaussianBlur, feMerge, feBlend, feComposite,
feColorMatrix matrix / saturate / hueRotate / luminanceToAlpha, feMorphology), text / tspan set in SVG fonts (font, font-face, glyph,
missing-glyph, hkern; ``fonts.py``), presentation attributes and ``style``.
Not supported (a warning, the element is skipped): textPath, image, foreignObject, switch, marker, ...
"""
from __future__ import annotations

import gzip
import io
import math
import os
import re
import warnings
import xml.etree.ElementTree as etree

import numpy as np

from .filters import COLOR_MATRIX_LUM, Filter, color_matrix_hue_rotate, color_matrix_saturate
from .fonts import FONT_STYLE_NORMAL, Font, FontsDB, Glyph
from .geometry import (
    PATH_CLOSED, PATH_FILL_NONZERO, PATH_LINE, STROKE_CAP_BUTT, STROKE_JOIN_MITER, Path, Transform,
)
from .layer import COMPOSE_ATOP, COMPOSE_IN, COMPOSE_OUT, COMPOSE_OVER, COMPOSE_XOR
from .paint import GradLinear, GradRadial, Pattern
from .scene import Scene

UNITS_USER = "userSpaceOnUse"
UNITS_BBOX = "objectBoundingBox"
FONT_SIZE = 12  # S:2658

# presentation attributes that children inherit (S:2729-2744)
_INHERITED = {
    "color", "fill", "fill-rule", "fill-opacity", "stroke", "stroke-opacity", "stroke-width", "stroke-linecap",
    "stroke-linejoin", "stroke-miterlimit", "font-family", "font-size", "font-weight", "text-anchor",
}
_NUMBER = re.compile(r"[-+]?(?:(?:\d*\.\d+)|(?:\d+\.?))(?:[Ee][+-]?\d+)?")
_HEX = re.compile("#?([0-9A-Fa-f]+)$")
_FUNC = re.compile(r"\s*(rgba?|hsl)\(([^\)]+)\)\s*")
_TRANSFORM_OP = re.compile(r"\s*(translate|scale|rotate|skewX|skewY|matrix)\s*\(([^\)]+)\)\s*")
_URL = re.compile(r"url\(\#([^)]+)\)")

# CSS named colours (the SVG 1.1 / CSS3 keyword table)
_NAMED = dict(zip(
    """aliceblue antiquewhite aqua aquamarine azure beige bisque black blanchedalmond blue blueviolet brown burlywood
    cadetblue chartreuse chocolate coral cornflowerblue cornsilk crimson cyan darkblue darkcyan darkgoldenrod darkgray
    darkgrey darkgreen darkkhaki darkmagenta darkolivegreen darkorange darkorchid darkred darksalmon darkseagreen
    darkslateblue darkslategray darkslategrey darkturquoise darkviolet deeppink deepskyblue dimgray dimgrey dodgerblue
    firebrick floralwhite forestgreen fuchsia gainsboro ghostwhite gold goldenrod gray grey green greenyellow honeydew
    hotpink indianred indigo ivory khaki lavender lavenderblush lawngreen lemonchiffon lightblue lightcoral lightcyan
    lightgoldenrodyellow lightgray lightgrey lightgreen lightpink lightsalmon lightseagreen lightskyblue lightslategray
    lightslategrey lightsteelblue lightyellow lime limegreen linen magenta maroon mediumaquamarine mediumblue
    mediumorchid mediumpurple mediumseagreen mediumslateblue mediumspringgreen mediumturquoise mediumvioletred
    midnightblue mintcream mistyrose moccasin navajowhite navy oldlace olive olivedrab orange orangered orchid
    palegoldenrod palegreen paleturquoise palevioletred papayawhip peachpuff peru pink plum powderblue purple
    rebeccapurple red rosybrown royalblue saddlebrown salmon sandybrown seagreen seashell sienna silver skyblue
    slateblue slategray slategrey snow springgreen steelblue tan teal thistle tomato turquoise violet wheat white
    whitesmoke yellow yellowgreen""".split(),
    """f0f8ff faebd7 00ffff 7fffd4 f0ffff f5f5dc ffe4c4 000000 ffebcd 0000ff 8a2be2 a52a2a deb887
    5f9ea0 7fff00 d2691e ff7f50 6495ed fff8dc dc143c 00ffff 00008b 008b8b b8860b a9a9a9
    a9a9a9 006400 bdb76b 8b008b 556b2f ff8c00 9932cc 8b0000 e9967a 8fbc8f
    483d8b 2f4f4f 2f4f4f 00ced1 9400d3 ff1493 00bfff 696969 696969 1e90ff
    b22222 fffaf0 228b22 ff00ff dcdcdc f8f8ff ffd700 daa520 808080 808080 008000 adff2f f0fff0
    ff69b4 cd5c5c 4b0082 fffff0 f0e68c e6e6fa fff0f5 7cfc00 fffacd add8e6 f08080 e0ffff
    fafad2 d3d3d3 d3d3d3 90ee90 ffb6c1 ffa07a 20b2aa 87cefa 778899
    778899 b0c4de ffffe0 00ff00 32cd32 faf0e6 ff00ff 800000 66cdaa 0000cd
    ba55d3 9370db 3cb371 7b68ee 00fa9a 48d1cc c71585
    191970 f5fffa ffe4e1 ffe4b5 ffdead 000080 fdf5e6 808000 6b8e23 ffa500 ff4500 da70d6
    eee8aa 98fb98 afeeee db7093 ffefd5 ffdab9 cd853f ffc0cb dda0dd b0e0e6 800080
    663399 ff0000 bc8f8f 4169e1 8b4513 fa8072 f4a460 2e8b57 fff5ee a0522d c0c0c0 87ceeb
    6a5acd 708090 708090 fffafa 00ff7f 4682b4 d2b48c 008080 d8bfd8 ff6347 40e0d0 ee82ee f5deb3 ffffff
    f5f5f5 ffff00 9acd32""".split(),
))
assert len(_NAMED) == 148


# ---------------------------------------------------------------------------------------------------------------------
# scalars
# ---------------------------------------------------------------------------------------------------------------------
def parse_float(text):
    """Number with an optional ``%`` (-> fraction) or ``px`` / ``pt`` suffix (S:3484-3495); None stays None."""
    if text is None or isinstance(text, float):
        return text
    text = text.strip()
    if text.endswith("%"):
        return float(text[:-1]) / 100.0
    if text.endswith("px") or text.endswith("pt"):
        return float(text[:-2])
    return float(text)


def parse_floats(text, at_least=None, at_most=None):
    if text is None:
        return None
    values = [float(v) for v in text.replace(",", " ").split(" ") if v]
    if at_least is not None and len(values) < at_least:
        raise ValueError(f"expected at least {at_least} arguments")
    if at_most is not None and len(values) > at_most:
        raise ValueError(f"expected at most {at_most} arguments")
    return values


def parse_angle(text) -> float:
    """Radians from ``<n>``, ``<n>deg`` (degrees) or ``<n>rad`` (S:3509-3516)."""
    text = text.strip()
    if text.endswith("deg"):
        return float(text[:-3]) * math.pi / 180
    if text.endswith("rad"):
        return float(text[:-3])
    return float(text) * math.pi / 180


def parse_size(text, default=None, dpi=96):
    """A length in user units (S:3519-3549): px, in, cm, mm, pt, pc, em, ex; ``%`` is not resolved."""
    if text is None:
        return default
    if isinstance(text, (int, float)):
        return float(text)
    text = text.strip().lower()
    m = _NUMBER.match(text)
    if m is None:
        warnings.warn(f"invalid size: {text}")
        return default
    value, unit = float(m.group(0)), text[m.end():].strip()
    if unit in ("", "px"):
        return value
    per_inch = {"in": 1, "cm": 2.54, "mm": 25.4, "pt": 72.0, "pc": 6.0}  # multiply by dpi first, then divide
    if unit in per_inch:
        return value * dpi if unit == "in" else value * dpi / per_inch[unit]
    if unit == "em":
        return value * FONT_SIZE
    if unit == "ex":
        return value * FONT_SIZE / 2.0
    if unit == "%":
        warnings.warn("size in % is not supported")
        return value
    return None


def parse_transform(text):
    """``transform`` attribute -> Transform, operations applied left to right (S:3416-3481); None stays None."""
    if text is None:
        return None
    tr = Transform()
    rest = text.strip().replace(",", " ")
    while rest:
        m = _TRANSFORM_OP.match(rest)
        if m is None:
            raise ValueError(f"failed to parse transform: {rest}")
        rest = rest[len(m.group(0)):]
        op, raw = m.groups()
        args = [a for a in raw.split(" ") if a]

        def need(*counts):
            if len(args) not in counts:
                raise ValueError(f"`{op}` transform requires {set(counts)} arguments {len(args)} where given")

        if op == "matrix":
            need(6)
            a, b, c, d, e, f = map(float, args)
            tr = tr.matrix(a, c, e, b, d, f)
        elif op == "translate":
            need(1, 2)
            v = list(map(float, args))
            tr = tr.translate(v[0], v[1] if len(v) == 2 else 0)
        elif op == "scale":
            need(1, 2)
            v = list(map(float, args))
            tr = tr.scale(v[0], v[1] if len(v) == 2 else v[0])
        elif op == "rotate":
            need(1, 3)
            angle = parse_angle(args[0])
            if len(args) == 1:
                tr = tr.rotate(angle)
            else:
                x, y = float(args[1]), float(args[2])
                tr = tr.translate(x, y).rotate(angle).translate(-x, -y)
        elif op == "skewX":
            need(1)
            tr = tr.skew(parse_angle(args[0]), 0)
        else:  # skewY
            need(1)
            tr = tr.skew(0, parse_angle(args[0]))
    return tr


def viewbox_transform(bbox, viewbox) -> Transform:
    """Fit `viewbox` into the viewport `bbox` = (x, y, w, h), uniform scale, centred (S:3116-3133)."""
    vx, vy, vw, vh = viewbox
    x, y, w, h = bbox
    if h is None and w is None:
        h, w = vh, vw
    elif h is None:
        h = vh * w / vw
    elif w is None:
        w = vw * h / vh
    scale = min(w / vw, h / vh)
    tx = -vx + (w / scale - vw) / 2 + x / scale
    ty = -vy + (h / scale - vh) / 2 + y / scale
    return Transform().scale(scale).translate(tx, ty)


# ---------------------------------------------------------------------------------------------------------------------
# colours and paints
# ---------------------------------------------------------------------------------------------------------------------
def _srgb_to_linear(rgba: np.ndarray) -> np.ndarray:
    rgb = rgba[:-1]
    small = rgb <= 0.04045
    rgb[small] = rgb[small] / 12.92
    large = ~small
    rgb[large] = np.power((rgb[large] + 0.055) / 1.055, 2.4)
    return rgba


def parse_color(text):
    """CSS colour -> premultiplied linear RGBA (S:3581-3624): #rgb[a], #rrggbb[aa], rgb()/rgba() with numbers or
    percentages, or a keyword; None (+ warning) when it is none of these."""
    color = None
    m = _HEX.match(text)
    if m is not None:
        digits = m.group(1)
        if len(digits) in (3, 4):
            color = np.array([int(c, 16) for c in digits], dtype=np.float64) / 15.0
        elif len(digits) in (6, 8):
            color = np.array([int(digits[i: i + 2], 16) for i in range(0, len(digits), 2)], dtype=np.float64) / 255.0
        else:
            raise ValueError(f"invalid hex color: {text}")
    m = _FUNC.match(text)
    if m is not None:
        kind, raw = m.groups()
        if kind.strip() not in ("rgb", "rgba"):
            raise ValueError(f"invalid rgb color: {text}")
        channels = []
        for ch in filter(None, raw.replace(",", " ").split(" ")):
            channels.append(float(ch[:-1]) / 100 if ch.endswith("%") else float(ch) / 255.0)
        color = np.array(channels)
    if color is not None:
        if color.shape == (3,):
            color = np.array([*color, 1.0], dtype=np.float64)
        color = _srgb_to_linear(color)
        color[:3] *= color[3:]
        return color
    named = _NAMED.get(text.lower().strip())
    if named is None:
        warnings.warn(f"invalid svg color: {text}")
        return None
    return parse_color("#" + named)


def _resolve_url(text, ids):
    m = _URL.match(text.strip())
    if m is None:
        return None
    target = ids.get(m.group(1))
    if target is None:
        warnings.warn(f"failed to resolve url: {text}")
    return target


def parse_paint(text, ids):
    """``fill`` / ``stroke`` value: none -> None, url(#id) -> the referenced paint, else a colour (S:3564-3578)."""
    if text is None:
        return None
    text = text.strip()
    if text == "none":
        return None
    target = _resolve_url(text, ids)
    if target is not None:
        return target
    color = parse_color(text)
    if color is None:
        warnings.warn(f"invalid paint: {text}")
    return color


def _expand_style(attrib, inherit=None) -> dict:
    """Element attributes with the ``style`` declarations folded in, on top of what the parent hands down (S:3103-3113)."""
    attrs = dict(attrib)
    style = attrs.pop("style", None)
    if style is not None:
        for decl in style.split(";"):
            if decl.strip():
                key, value = decl.split(":", 1)
                attrs[key.strip()] = value.strip()
    return attrs if inherit is None else {**inherit, **attrs}


def _gradient_stops(element):
    stops = []
    for child in element:
        if not child.tag.endswith("stop"):
            continue
        attrs = _expand_style(child.attrib)
        offset = parse_float(attrs.get("offset")) or 0
        offset = min(max(offset, 0), 1)
        color = parse_color(attrs["stop-color"])
        if color is None:
            continue
        opacity = attrs.get("stop-opacity")
        if opacity:
            color *= float(opacity)
        stops.append((offset, color))
    stops.sort(key=lambda s: s[0])
    return stops


def _gradient(element, linear: bool):
    """<linearGradient> / <radialGradient> -> GradLinear / GradRadial, a plain colour (one stop) or None (no stops).
    Like the reference (S:2873-2878, S:3181-3249) a gradient is built from its own element only: ``href`` is not followed."""
    attr = element.attrib
    text = attr.get("gradientTransform") or attr.get("transform")
    transform = parse_transform(text) if text is not None else None
    spread = attr.get("spreadMethod", "pad")
    units = attr.get("gradientUnits", UNITS_BBOX)
    if units not in (UNITS_BBOX, UNITS_USER):
        raise ValueError(f"invalid gradient unites: {units}")
    bbox_units = units == UNITS_BBOX
    stops = _gradient_stops(element)
    if not stops:
        return None
    if len(stops) == 1:
        return stops[0][1]
    interp = attr.get("color-interpolation")
    linear_rgb = True if interp == "linearRGB" else (False if interp == "sRGB" else None)
    if linear:
        p0 = np.array([parse_float(attr.get("x1", "0")), parse_float(attr.get("y1", "0"))])
        p1 = np.array([parse_float(attr.get("x2", "1")), parse_float(attr.get("y2", "0"))])
        return GradLinear(p0, p1, stops, transform, spread, bbox_units, linear_rgb)
    cx, cy = parse_float(attr.get("cx", "0.5")), parse_float(attr.get("cy", "0.5"))
    fx, fy = parse_float(attr.get("fx")), parse_float(attr.get("fy"))
    fcenter = None
    if fx is not None or fy is not None:
        fcenter = np.array([cx if fx is None else fx, cy if fy is None else fy])
    radius = parse_float(attr.get("r")) or 0.5
    return GradRadial(np.array([cx, cy]), radius, fcenter, parse_float(attr.get("fr")), stops, transform, spread,
                      bbox_units, linear_rgb)


_COMPOSITE_OPERATORS = {"over": COMPOSE_OVER, "in": COMPOSE_IN, "out": COMPOSE_OUT, "atop": COMPOSE_ATOP, "xor": COMPOSE_XOR}


def _filter(element) -> Filter:
    """<filter> -> Filter chain (S:3271-3362)."""
    flt = Filter.empty()
    for child in element:
        tag = child.tag.split("}")[-1]
        attrs = child.attrib
        result, src = attrs.get("result"), attrs.get("in")
        if tag == "feOffset":
            flt = flt.offset(parse_float(attrs.get("dx", "0")), parse_float(attrs.get("dy", "0")), src, result)
        elif tag == "feGaussianBlur":
            stds = parse_floats(attrs.get("stdDeviation"), 1, 2)
            if stds is not None:
                std_x, std_y = stds * 2 if len(stds) == 1 else stds
                flt = flt.blur(std_x, std_y, src, result)
        elif tag == "feMerge":
            flt = flt.merge([n.get("in") for n in child if n.tag.split("}")[-1] == "feMergeNode"], result)
        elif tag == "feBlend":
            flt = flt.blend(src, attrs.get("in2"), attrs.get("mode"), result)
        elif tag == "feComposite":
            op = attrs.get("operator", "over")
            if op == "arithmetic":
                mode = tuple(parse_float(attrs.get(k, "0")) for k in ("k1", "k2", "k3", "k4"))
            elif op in _COMPOSITE_OPERATORS:
                mode = _COMPOSITE_OPERATORS[op]
            else:
                warnings.warn(f"unsupported composite mode: {op}")
                mode = COMPOSE_OVER
            flt = flt.composite(src, attrs.get("in2"), mode, result)
        elif tag == "feColorMatrix":
            kind, values = attrs.get("type", "matrix"), attrs.get("values")
            if kind == "matrix":
                matrix = np.eye(4, 5) if values is None else np.array(parse_floats(values, 20, 20)).reshape(4, 5)
            elif kind == "saturate":
                matrix = color_matrix_saturate(1 if values is None else parse_float(values))
            elif kind == "hueRotate":
                matrix = color_matrix_hue_rotate(0 if values is None else parse_angle(values))
            elif kind == "luminanceToAlpha":
                matrix = COLOR_MATRIX_LUM
            else:
                warnings.warn(f"unsupported color matrix type: {kind}")
                matrix = None
            if matrix is not None:
                flt = flt.color_matrix(src, matrix, result)
        elif tag == "feMorphology":
            method = {"erode": "min", "dilate": "max"}.get(attrs.get("operator", "erode"))
            if method is None:
                warnings.warn(f"invalid morphology operator: {attrs.get('operator')}")
            radius = parse_floats(attrs.get("radius", "0"), 1, 2)
            rx, ry = (radius[0], radius[0]) if len(radius) == 1 else radius
            if method is not None and rx > 0 and ry > 0:
                flt = flt.morphology(rx, ry, method, src, result)
        else:
            warnings.warn(f"unsupported filter type: {tag}")
    return flt


def _font_weight(text) -> int:
    if text is None:
        return 400
    text = text.lower()
    return {"normal": 400, "bold": 700}.get(text) or int(float(text))


def _names_to_unicode(names, by_name) -> list:
    """Glyph names of an hkern ``g1`` / ``g2`` list -> their unicode strings (unknown or unicode-less names drop out)."""
    out = []
    for name in filter(None, (names or "").split(",")):
        glyph = by_name.get(name)
        if glyph is not None and glyph.unicode:
            out.append(glyph.unicode)
    return out


def _font(element):
    """<font> -> Font, or None without a <font-face> (S:3627-3702).  Children inherit the <font>'s own attributes, so a
    ``horiz-adv-x`` on the <font> is the default advance; a glyph lacking ``unicode`` or any advance is dropped; kerning
    pairs are the cross product of (u1 + g1) x (u2 + g2), later <hkern> elements overriding earlier ones."""
    glyphs, by_name, kerning = {}, {}, {}
    missing, font = None, None
    for child in element:
        tag = child.tag.split("}")[-1]
        attrs = _expand_style(child.attrib, element.attrib)
        if tag == "glyph":
            code, advance = attrs.get("unicode"), attrs.get("horiz-adv-x")
            if code is None or advance is None:
                continue
            glyph = Glyph(code, float(advance), attrs.get("d", ""), attrs.get("glyph-name"))
            glyphs[code] = glyph
            if glyph.name is not None:
                by_name[glyph.name] = glyph
        elif tag == "missing-glyph":
            missing = Glyph(None, float(attrs.get("horiz-adv-x")), attrs.get("d", ""), "missing-glyph")
        elif tag == "font-face":
            upm = float(attrs.get("units-per-em", "2048"))
            font = Font(attrs.get("font-family", f"{id(element)}"), _font_weight(attrs.get("font-weight")),
                        attrs.get("font-style", FONT_STYLE_NORMAL), float(attrs.get("ascent", str(upm))),
                        float(attrs.get("descent", "0")), upm)
        elif tag == "hkern":
            left = list(filter(None, (attrs.get("u1") or "").split(","))) + _names_to_unicode(attrs.get("g1"), by_name)
            right = list(filter(None, (attrs.get("u2") or "").split(","))) + _names_to_unicode(attrs.get("g2"), by_name)
            if attrs.get("k") is None:
                continue
            for a in left:
                for b in right:
                    kerning[(a, b)] = float(attrs["k"])
    if font is None:
        warnings.warn("font is missing `font-face` element")
        return None
    font.glyphs.update(glyphs)
    font.hkern.update(kerning)
    if missing is not None:
        font.missing_glyph = missing
    return font


# ---------------------------------------------------------------------------------------------------------------------
# shapes -> path data (through the same %g-formatted strings the reference builds, S:3365-3413)
# ---------------------------------------------------------------------------------------------------------------------
def rect_path_data(x, y, width, height, rx=None, ry=None) -> str:
    if rx is None or ry is None:
        r = rx if rx is not None else ry
        rx, ry = (r, r) if r is not None else (0, 0)
    rounded = rx > 0 and ry > 0
    d = [f"M{x + rx:g},{y:g}", f"H{x + width - rx:g}"]
    if rounded:
        d.append(f"A{rx:g},{ry:g},0,0,1,{x + width:g},{y + ry:g}")
    d.append(f"V{y + height - ry:g}")
    if rounded:
        d.append(f"A{rx:g},{ry:g},0,0,1,{x + width - rx:g},{y + height:g}")
    d.append(f"H{x + rx:g}")
    if rounded:
        d.append(f"A{rx:g},{ry:g},0,0,1,{x:g},{y + height - ry:g}")
    d.append(f"V{y + ry:g}")
    if rounded:
        d.append(f"A{rx:g},{ry:g},0,0,1,{x + rx:g},{y:g}")
    d.append("z")
    return " ".join(d)


def ellipse_path_data(cx, cy, rx, ry) -> str:
    if rx is None or ry is None:
        r = rx if rx is not None else ry
        if r is None:
            return ""
        rx = ry = r
    return " ".join([
        f"M{cx + rx:g},{cy:g}", f"A{rx:g},{ry:g},0,0,1,{cx:g},{cy + ry:g}", f"A{rx:g},{ry:g},0,0,1,{cx - rx:g},{cy:g}",
        f"A{rx:g},{ry:g},0,0,1,{cx:g},{cy - ry:g}", f"A{rx:g},{ry:g},0,0,1,{cx + rx:g},{cy:g}", "z",
    ])


# ---------------------------------------------------------------------------------------------------------------------
# the loader
# ---------------------------------------------------------------------------------------------------------------------
class _Loader:
    def __init__(self, fg, width, fonts=None):
        self.fonts = FontsDB() if fonts is None else fonts
        self.ids: dict = {}
        self.size = None
        self.fg = fg
        self.width = width

    # -- leaves --------------------------------------------------------------------------------------------------------
    def shape(self, attrs, path=None) -> list:
        """Fill and stroke scenes of one shape from its (inherited + own) attributes (S:3136-3178)."""
        if path is None:
            d = attrs.get("d")
            if d is None:
                return []
            path = Path.from_svg(d)
        out = []
        fill = attrs.get("fill")
        if fill is not None:
            fill = attrs.get("color") if fill == "currentColor" else parse_paint(fill, self.ids)
        elif self.fg is not None:
            fill = self.fg
        else:
            fill = np.array([0, 0, 0, 1], dtype=np.float64)
        if fill is not None:
            node = Scene.fill(path, fill, attrs.get("fill-rule", PATH_FILL_NONZERO))
            opacity = parse_float(attrs.get("fill-opacity"))
            out.append(node if opacity is None else node.opacity(opacity))
        stroke = attrs.get("stroke")
        stroke = attrs.get("color") if stroke == "currentColor" else parse_paint(stroke, self.ids)
        if stroke is not None:
            node = Scene.stroke(path, stroke, parse_float(attrs.get("stroke-width", "1")), attrs.get("stroke-linecap"),
                                attrs.get("stroke-linejoin"))
            opacity = parse_float(attrs.get("stroke-opacity"))
            out.append(node if opacity is None else node.opacity(opacity))
        return out

    def text(self, element, attrs) -> list:
        """<text> with nested <tspan>: one transformed shape per run of characters (S:3716-3788).

        The pen starts at (0, 0); ``x`` / ``y`` set it, ``dx`` / ``dy`` move it, and each run advances it by its set width.
        White space collapses to single blanks; a run keeps one leading blank unless the previous run ended in one, and
        one trailing blank.  ``text-anchor`` shifts the whole element by its total advance measured from the ``x`` of
        the <text> itself.
        """
        def run(text, attrs, pen, after_blank):
            ox, oy = pen
            for key in ("x", "dx", "y", "dy"):  # consumed here, so that they do not reach the runs that follow
                value = parse_size(attrs.pop(key, None))
                if value is None:
                    continue
                if key == "x":
                    ox = value
                elif key == "y":
                    oy = value
                elif key == "dx":
                    ox += value
                else:
                    oy += value
            if not text:
                return [], (ox, oy), after_blank
            text = text.replace("\n", " ")
            lead = " " if text[0] in " \t" and len(text) > 1 and not after_blank else ""
            trail = " " if text[-1] in " \t" else ""
            words = " ".join(text.split())
            if not words:
                return [], (ox, oy), after_blank
            words = lead + words + trail
            size = parse_float(attrs.get("font-size", f"{FONT_SIZE}"))
            font = self.fonts.resolve(attrs.get("font-family"), _font_weight(attrs.get("font-weight")))
            if font is None:
                return [], (ox, oy), after_blank
            path, advance = font.str_to_path(size, words)
            place = Transform().translate(ox, oy)
            return [node.transform(place) for node in self.shape(attrs, path)], (ox + advance, oy), bool(trail)

        def walk(element, attrs, pen, after_blank):
            out, pen, after_blank = run(element.text, attrs, pen, after_blank)
            # whether children are descended into is decided by the tag of ``element`` itself (S:3766-3767)
            descend = element.tag.split("}")[-1] in ("text", "tspan")
            for child in element:
                if descend:
                    nodes, pen, after_blank = walk(child, _expand_style(child.attrib, attrs), pen, after_blank)
                    out.extend(nodes)
                nodes, pen, after_blank = run(child.tail, attrs, pen, after_blank)
                out.extend(nodes)
            return out, pen, after_blank

        start_x = parse_float(attrs.get("x", "0"))
        nodes, (end_x, _), _ = walk(element, attrs, (0, 0), True)
        anchor = attrs.get("text-anchor")
        if anchor in ("middle", "end"):
            shift = Transform().translate((start_x - end_x) / (2 if anchor == "middle" else 1), 0)
            nodes = [node.transform(shift) for node in nodes]
        return nodes

    def children(self, element, inherit) -> list:
        out = []
        for child in element:
            out.extend(self.element(child, inherit))
        return out

    # -- containers ----------------------------------------------------------------------------------------------------
    def svg(self, element, attrs, inherit, top) -> list:
        group = self.children(element, inherit)
        if not group:
            return []
        scene = Scene.group(group)
        x, y = parse_size(attrs.get("x", "0")), parse_size(attrs.get("y", "0"))
        w, h = parse_size(attrs.get("width")), parse_size(attrs.get("height"))
        viewbox = [0, 0, w, h] if w is not None and h is not None else None
        width = self.width if top else None
        if width is not None:
            w, h = (width, int(width * h / w)) if w is not None and h is not None else (width, None)
        viewbox = parse_floats(attrs.get("viewBox"), 4, 4) or viewbox
        if viewbox is not None:
            scene = scene.transform(viewbox_transform((x, y, w, h), viewbox))
            _vx, _vy, vw, vh = viewbox
            if h is None and w is None:
                h, w = vh, vw
            elif h is None:
                h = vh * w / vw
            elif w is None:
                w = vw * h / vh
        elif x > 0 and y > 0:
            scene = scene.transform(Transform().translate(x, y))
        if w is not None and h is not None:
            if top:
                self.size = (w, h)
            else:  # a nested viewport clips its content
                frame = [(PATH_LINE, [[x, y], [x + w, y]]), (PATH_LINE, [[x + w, y], [x + w, y + h]]),
                         (PATH_LINE, [[x + w, y + h], [x, y + h]]), (PATH_CLOSED, [[x, y + h], [x, y]])]
                scene = scene.clip(Scene.fill(Path([frame]), np.ones(4)))
        return [scene]

    def element(self, element, inherit, top=False) -> list:
        tag = element.tag.split("}")[-1]
        attrs = _expand_style(element.attrib, inherit)
        inherit = {k: v for k, v in attrs.items() if k in _INHERITED}
        ids = self.ids
        group: list = []
        if tag == "svg":
            group = self.svg(element, attrs, inherit, top)
        elif tag == "path":
            group = self.shape(attrs)
        elif tag == "g":
            group = self.children(element, inherit)
        elif tag == "defs":
            self.children(element, inherit)
        elif tag in ("linearGradient", "radialGradient"):
            if attrs.get("id") is not None:
                ids[attrs["id"]] = _gradient(element, tag == "linearGradient")
            return []
        elif tag == "clipPath":
            inherit.setdefault("fill-rule", attrs.get("clip-rule"))
            if attrs.get("id") is not None:
                content = self.children(element, inherit)
                if content:
                    scene = Scene.group(content)
                    tr = parse_transform(attrs.get("transform"))
                    if tr is not None:
                        scene = scene.transform(tr)
                    ids[attrs["id"]] = (scene, attrs.get("clipPathUnits") == UNITS_BBOX)
            return []
        elif tag == "mask":
            if attrs.get("id") is not None:
                scene = Scene.group(self.children(element, inherit))
                tr = parse_transform(attrs.get("transform"))
                if tr is not None:
                    scene = scene.transform(tr)
                ids[attrs["id"]] = (scene, attrs.get("maskContentUnits") == UNITS_BBOX)
        elif tag == "filter":
            if attrs.get("id") is not None:
                ids[attrs["id"]] = _filter(element)
        elif tag == "pattern":  # S:2914-2951
            if attrs.get("id") is not None:
                w, h = parse_float(attrs.get("width")), parse_float(attrs.get("height"))
                if w is None or h is None:
                    return []
                content = Scene.group(self.children(element, inherit))
                tr = parse_transform(attrs.get("patternTransform"))
                ids[attrs["id"]] = Pattern(
                    content, attrs.get("patternContentUnits", UNITS_USER) == UNITS_BBOX,
                    parse_floats(attrs.get("viewBox"), 4, 4), parse_float(attrs.get("x", "0")), parse_float(attrs.get("y", "0")),
                    w, h, Transform() if tr is None else tr, attrs.get("patternUnits", UNITS_BBOX) == UNITS_BBOX)
        elif tag == "rect":
            x, y = parse_size(attrs.pop("x", "0")), parse_size(attrs.pop("y", "0"))
            w, h = parse_size(attrs.pop("width")), parse_size(attrs.pop("height"))
            attrs["d"] = rect_path_data(x, y, w, h, parse_size(attrs.get("rx")), parse_size(attrs.get("ry")))
            group = self.shape(attrs)
        elif tag == "circle":
            cx, cy = parse_size(attrs.pop("cx", "0")), parse_size(attrs.pop("cy", "0"))
            r = parse_size(attrs.pop("r"))
            attrs["d"] = ellipse_path_data(cx, cy, r, r)
            group = self.shape(attrs)
        elif tag == "ellipse":
            cx, cy = parse_size(attrs.pop("cx", "0")), parse_size(attrs.pop("cy", "0"))
            attrs["d"] = ellipse_path_data(cx, cy, parse_size(attrs.pop("rx")), parse_size(attrs.pop("ry")))
            group = self.shape(attrs)
        elif tag == "polygon":
            attrs["d"] = f"M{attrs.pop('points')}z"
            group = self.shape(attrs)
        elif tag == "polyline":
            attrs["d"] = f"M{attrs.pop('points')}"
            group = self.shape(attrs)
        elif tag == "line":
            x1, y1 = parse_size(attrs.pop("x1", "0")), parse_size(attrs.pop("y1", "0"))
            x2, y2 = parse_size(attrs.pop("x2", "0")), parse_size(attrs.pop("y2", "0"))
            attrs["d"] = f"M{x1},{y1} {x2},{y2}"
            group = self.shape(attrs)
        elif tag in ("title", "desc", "metadata"):
            return []
        elif tag == "font":
            font = _font(element)
            if font is not None:
                self.fonts.register(font, attrs.get("id"))
                if attrs.get("id") is not None:
                    ids[attrs["id"]] = font
            return []
        elif tag == "text":
            group = self.text(element, attrs)
        elif tag == "use":
            x, y = attrs.get("x"), attrs.get("y")
            if x is not None or y is not None:
                attrs["transform"] = attrs.get("transform", "") + f" translate({x}, {y})"
            href = attrs.get("href")
            if href is None:
                href = next((v for k, v in attrs.items() if k.endswith("}href")), None)
            if href and href.startswith("#"):
                item = ids.get(href[1:])
                if isinstance(item, Scene):
                    group = [item]
        else:
            warnings.warn(f"unsupported element type: {tag}")

        if not group:
            return group
        # decorations, innermost first: filter, group opacity, clip-path, mask; the element's transform goes last, so that
        # clips and masks live in the transformed space (S:3031-3071)
        name = attrs.get("filter")
        if name is not None:
            flt = _resolve_url(name, ids)
            if isinstance(flt, Filter):
                group = [Scene.group(group).filter(flt)]
            else:
                warnings.warn(f"not a filter referenced {name}: {type(flt)}")
        opacity = parse_float(attrs.get("opacity"))
        if opacity is not None:
            group = [Scene.group(group).opacity(opacity)]
        for key, wrap in (("clip-path", "clip"), ("mask", "mask")):
            ref = attrs.get(key)
            if ref is None:
                continue
            target = _resolve_url(ref, ids)
            if isinstance(target, tuple):
                scene, bbox_units = target
                group = [getattr(Scene.group(group), wrap)(scene, bbox_units)]
            else:
                warnings.warn(f"{key} expected {ref}: {type(target)}")
        tr = parse_transform(attrs.get("transform"))
        if tr is not None:
            group = [node.transform(tr) for node in group]
        if attrs.get("id") is not None:
            ids[attrs["id"]] = Scene.group(group)
        return group


def svg_scene(file, fg=None, width=None, fonts=None):
    """Load an SVG document from a file object: ``(Scene | None, ids, size)`` with ``size = (width, height)`` of the
    outermost viewport (S:2803-3083).  ``width`` rescales the document to that many pixels, ``fg`` replaces the default
    black of shapes without a ``fill``, ``fonts`` is the ``FontsDB`` text is set from (<font> elements of the document are
    added to it)."""
    loader = _Loader(fg, width, fonts)
    root = etree.parse(file).getroot()
    inherit = dict(color=np.array([0.0, 0.0, 0.0, 1.0]) if fg is None else fg)
    group = loader.element(root, inherit, top=True)
    if not group:
        return None, loader.ids, loader.size
    return Scene.group(group), loader.ids, loader.size


def svg_scene_from_str(text: str, fg=None, width=None, fonts=None):
    return svg_scene(io.StringIO(text), fg, width, fonts)


def svg_scene_from_filepath(path: str, fg=None, width=None, fonts=None):
    path = os.path.expanduser(path)
    if os.path.splitext(path)[1] in (".gz", ".svgz"):
        with gzip.open(path, mode="rt", encoding="utf-8") as f:
            return svg_scene(f, fg, width, fonts)
    with open(path, encoding="utf-8") as f:
        return svg_scene(f, fg, width, fonts)


def render_svg(svg, output=None, bg=None, fg=None, width=None, id=None, transform=None, linear_rgb=False, fonts=None,
               level: int = 9, threads: int = 1):
    """Document in, PNG out: the steps of the reference's command line (S:3796-3877) as one library call, every pixel
    operation on the device.  ``svg`` is a file path or a file object; ``bg`` / ``fg`` are colours as ``parse_color``
    returns them; ``id`` renders a single element (on its own bounding box); ``transform`` is applied on top of the x/y
    swap of presentation space; ``level`` / ``threads`` go to the PNG writer (the defaults write the reference's exact
    file, ``threads`` > 1 the same pixels much faster).  Returns the PNG bytes (also written to ``output``: a path or a binary file object)
    or None when there is nothing to draw."""
    view = Transform().matrix(0, 1, 0, 1, 0, 0)
    if transform is not None:
        view = view @ transform
    if isinstance(svg, (str, os.PathLike)):
        scene, ids, size = svg_scene_from_filepath(os.fspath(svg), fg=fg, width=width, fonts=fonts)
    else:
        scene, ids, size = svg_scene(svg, fg=fg, width=width, fonts=fonts)
    if scene is not None and id is not None:
        scene, size = ids.get(id), None
        if not isinstance(scene, Scene):
            raise KeyError(f"no object with id: {id}")
    if scene is None:
        return None
    if size is not None:
        w, h = size
        result = scene.render(view, viewport=[0, 0, int(h), int(w)], linear_rgb=linear_rgb)
    else:
        result = scene.render(view, linear_rgb=linear_rgb)
    if result is None:
        return None
    layer, _hull = result
    if size is not None:
        layer = layer.convert(pre_alpha=True, linear_rgb=linear_rgb).on_canvas(int(h), int(w))
    if bg is not None:
        layer = layer.background(bg)
    png = layer.write_png(None, level, threads).getvalue()
    if isinstance(output, (str, os.PathLike)):
        with open(output, "wb") as f:
            f.write(png)
    elif output is not None:
        output.write(png)
    return png
