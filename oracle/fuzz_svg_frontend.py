#!/usr/bin/env python3
"""Randomised cross-check of the SVG front-end against the reference's loader (TEST INFRASTRUCTURE; build container only).

Generates documents from a small grammar -- path data with every command in random number formats, basic shapes,
nested groups with transforms / style / opacity, gradients, clip paths, masks, patterns, <use>, nested <svg> -- loads
each with the reference's svg_scene (S:2803) and with svgrasterize_amd.svg, and compares the scene dumps.  A document
the reference rejects (exception) must be rejected here as well.

    python oracle/fuzz_svg_frontend.py [n_documents] [first_seed]
"""
import json
import os
import random
import sys
import warnings

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))
import gen_golden  # noqa: E402
import ref_loader  # noqa: E402


class TooSlow(Exception):
    pass


class time_limit:
    """``with time_limit(s):`` raises TooSlow after s seconds (the reference's stroker does not terminate on some inputs)."""

    def __init__(self, seconds: int):
        self.seconds = seconds

    def __enter__(self):
        import signal

        def fire(*_a):
            raise TooSlow()

        self.old = signal.signal(signal.SIGALRM, fire)
        signal.alarm(self.seconds)

    def __exit__(self, *exc):
        import signal

        signal.alarm(0)
        signal.signal(signal.SIGALRM, self.old)
        return False


DEGENERATE = os.environ.get("FUZZ_DEGENERATE") == "1"  # now and then a zero, a tiny or a repeated value: empty shapes, cusps, ...


def num(r, lo=-60.0, hi=160.0):
    v = r.uniform(lo, hi)
    if DEGENERATE and r.random() < 0.12:
        v = r.choice([0.0, 0.0, 1e-9, -1e-9, 1.0, lo, hi, round(v), 0.5])
    style = r.randrange(6)
    if style == 0:
        return str(int(v))
    if style == 1:
        return f"{v:.1f}"
    if style == 2:
        return f"{v:.4g}"
    if style == 3:
        return f"{v:e}"
    if style == 4:
        return (f"{v:.3f}").replace("0.", ".") if abs(v) < 1 else f"{v:.3f}"
    return repr(v)


def sep(r):
    return r.choice([",", " ", " , ", "\n", "\t "])


def path_data(r):
    out = [r.choice("Mm") + num(r) + sep(r) + num(r)]
    for _ in range(r.randrange(1, 14)):
        c = r.choice("LlHhVvCcSsQqTtAaZzMm")
        if c in "Zz":
            out.append(c)
            continue
        n = dict(L=2, H=1, V=1, C=6, S=4, Q=4, T=2, A=7, M=2)[c.upper()]
        reps = r.choice([1, 1, 1, 2, 3])
        args = []
        for _k in range(reps):
            if c in "Aa":
                args += [num(r, 0, 40), num(r, 0, 40), num(r, -180, 180), r.choice("01"), r.choice("01"), num(r), num(r)]
            else:
                args += [num(r, -30, 30) if c.islower() else num(r) for _j in range(n)]
        text = c
        for k, a in enumerate(args):
            text += ("" if k == 0 and r.random() < 0.5 else sep(r)) + a
        out.append(text)
    return r.choice([" ", "", "\n"]).join(out)


COLORS = ["red", "#0a3", "#12345678", "rgb(10, 200, 30)", "rgba(10%, 20%, 30%, 0.5)", "none", "black", "#fc0", "#FFf",
          "currentColor", "url(#g0)", "url(#g1)", "url(#p0)", "url(#nope)", "tomato", "  navy "]


def presentation(r):
    attrs = {}
    if r.random() < 0.7:
        attrs["fill"] = r.choice(COLORS)
    if r.random() < 0.4:
        attrs["stroke"] = r.choice(COLORS)
        if r.random() < 0.7:
            attrs["stroke-width"] = num(r, 0.2, 6)
        if r.random() < 0.4:
            attrs["stroke-linecap"] = r.choice(["butt", "round", "square"])
        if r.random() < 0.4:
            attrs["stroke-linejoin"] = r.choice(["miter", "round", "bevel"])
    for key, p in (("fill-opacity", 0.2), ("stroke-opacity", 0.15), ("opacity", 0.2)):
        if r.random() < p:
            attrs[key] = f"{r.uniform(0.1, 1.0):.2f}"
    if r.random() < 0.2:
        attrs["fill-rule"] = r.choice(["nonzero", "evenodd"])
    if r.random() < 0.25:
        ops = []
        for _ in range(r.randrange(1, 3)):
            k = r.randrange(6)
            ops.append([f"translate({num(r, -20, 20)} {num(r, -20, 20)})", f"scale({num(r, 0.5, 2)})", f"scale({num(r, 0.5, 2)}, {num(r, 0.5, 2)})",
                        f"rotate({num(r, -90, 90)})", f"rotate({num(r, -90, 90)} {num(r, 0, 50)} {num(r, 0, 50)})",
                        f"matrix({num(r, 0.5, 1.5)} {num(r, -0.3, 0.3)} {num(r, -0.3, 0.3)} {num(r, 0.5, 1.5)} {num(r, -10, 10)} {num(r, -10, 10)})",
                        ][k])
        attrs["transform"] = " ".join(ops)
    if r.random() < 0.1:
        attrs["clip-path"] = r.choice(["url(#c0)", "url(#c1)"])
    if r.random() < 0.05:
        attrs["mask"] = "url(#m0)"
    if r.random() < (0.3 if BIG_FILTERS else 0.07):
        attrs["filter"] = r.choice(["url(#f0)", "url(#f1)"])
    if r.random() < 0.3 and attrs:  # move some of them into a style attribute
        keys = r.sample(sorted(k for k in attrs if k != "transform"), k=min(2, len([k for k in attrs if k != "transform"])))
        if keys:
            attrs["style"] = "; ".join(f"{k}: {attrs.pop(k)}" for k in keys) + r.choice(["", ";"])
    return "".join(f' {k}="{v}"' for k, v in attrs.items())


FONT = ('<font id="fz" horiz-adv-x="520"><font-face font-family="Fuzz" units-per-em="1000" ascent="800" descent="-200"/>'
        '<missing-glyph horiz-adv-x="400" d="M40,0 H360 V600 H40 z"/>'
        '<glyph unicode="a" glyph-name="a" d="M0,0 L200,600 L400,0 H320 L270,150 H130 L80,0 z"/>'
        '<glyph unicode="b" glyph-name="b" horiz-adv-x="480" d="M60,0 V700 H300 Q420,700 420,540 T300,380 Q440,380 440,190 T300,0 z"/>'
        '<glyph unicode="c" glyph-name="c" horiz-adv-x="450" d="M400,80 C300,-40 40,-20 40,250 S300,540 400,420 L340,370 C280,450 120,430 120,250 S280,50 340,130 z"/>'
        '<glyph unicode="ab" glyph-name="ab" horiz-adv-x="800" d="M0,0 L200,600 L400,0 H760 V700 H680 V80 H340 z"/>'
        '<glyph unicode=" " glyph-name="sp" horiz-adv-x="260" d=""/>'
        '<hkern u1="a" u2="c" k="60"/><hkern g1="b" g2="a,c" k="-35"/></font>')


def text_element(r, p):
    def words():
        return "".join(r.choice(["a", "b", "c", "ab", " ", "  ", "x", "\n", "a c"]) for _ in range(r.randrange(1, 7)))

    inner = words()
    for _ in range(r.randrange(0, 3)):
        attrs = "".join(f' {k}="{num(r, -8, 8)}"' for k in ("dx", "dy") if r.random() < 0.4)
        if r.random() < 0.3:
            attrs += f' x="{num(r, 0, 90)}"'
        if r.random() < 0.3:
            attrs += f' fill="{r.choice(COLORS[:9])}"'
        if r.random() < 0.2:
            attrs += f' font-size="{num(r, 6, 30)}"'
        inner += f"<tspan{attrs}>{words()}</tspan>" + (words() if r.random() < 0.5 else "")
    anchor = r.choice(["", "", ' text-anchor="middle"', ' text-anchor="end"'])
    family = r.choice(["Fuzz", "fuzz", "Nothing Sans", "Fuzz"])
    return f'<text x="{num(r, 0, 90)}" y="{num(r, 10, 90)}" font-family="{family}" font-size="{num(r, 8, 36)}"{anchor}{p}>{inner}</text>'


def shape(r):
    k = r.randrange(9)
    p = presentation(r)
    if k == 8:
        return text_element(r, p)
    if k == 0:
        return f'<rect x="{num(r, 0, 80)}" y="{num(r, 0, 80)}" width="{num(r, 1, 60)}" height="{num(r, 1, 60)}"' + \
               (f' rx="{num(r, 0, 9)}"' if r.random() < 0.4 else "") + (f' ry="{num(r, 0, 9)}"' if r.random() < 0.3 else "") + p + "/>"
    if k == 1:
        return f'<circle cx="{num(r, 0, 100)}" cy="{num(r, 0, 100)}" r="{num(r, 1, 40)}"{p}/>'
    if k == 2:
        return f'<ellipse cx="{num(r, 0, 100)}" cy="{num(r, 0, 100)}" rx="{num(r, 1, 40)}" ry="{num(r, 1, 30)}"{p}/>'
    if k == 3:
        return f'<line x1="{num(r)}" y1="{num(r)}" x2="{num(r)}" y2="{num(r)}"{p}/>'
    if k == 4:
        pts = " ".join(f"{num(r, 0, 120)},{num(r, 0, 120)}" for _ in range(r.randrange(3, 7)))
        return f'<{r.choice(["polygon", "polyline"])} points="{pts}"{p}/>'
    if k == 5:
        return f'<use href="#sym" x="{num(r, 0, 60)}" y="{num(r, 0, 60)}"{p}/>'
    return f'<path d="{path_data(r)}"{p}/>'


MANY = os.environ.get("FUZZ_MANY") == "1"  # dozens of shapes per group: long tile lists, many masks per pre-pass


def group(r, depth):
    body = []
    for _ in range(r.randrange(12, 40) if MANY and depth == 0 else r.randrange(1, 5)):
        if depth < 3 and r.random() < 0.25:
            body.append(group(r, depth + 1))
        elif depth < 2 and r.random() < 0.08:
            body.append(f'<svg x="{num(r, 0, 50)}" y="{num(r, 0, 50)}" width="{num(r, 20, 90)}" height="{num(r, 20, 90)}"' +
                        (f' viewBox="0 0 {num(r, 20, 200)} {num(r, 20, 200)}"' if r.random() < 0.6 else "") + ">" + shape(r) + shape(r) + "</svg>")
        else:
            body.append(shape(r))
    return f"<g{presentation(r)}>" + "".join(body) + "</g>"


def stops(r):
    return "".join(f'<stop offset="{r.choice([num(r, 0, 1), str(r.randrange(0, 101)) + "%"])}" stop-color="{r.choice(COLORS[:9])}"' +
                   (f' stop-opacity="{r.uniform(0.1, 1):.2f}"' if r.random() < 0.3 else "") + "/>" for _ in range(r.choice([0, 1, 2, 2, 3, 3, 4, 6, 9])))


SPREADS = ["", ' spreadMethod="reflect"', ' spreadMethod="repeat"']
CLIP_RULES = ["", ' clip-rule="evenodd"']


BIG_FILTERS = os.environ.get("FUZZ_BIGFILTER") == "1"  # blurs, offsets and morphology radii as large as the shapes


def filters(r):
    s_hi, o_hi, m_hi = (25, 40, 12) if BIG_FILTERS else (3, 4, 2.5)
    shadow = (f'<filter id="f0"><feGaussianBlur in="SourceAlpha" stdDeviation="{num(r, 0.4, s_hi)}' + (f' {num(r, 0.4, s_hi)}' if r.random() < 0.4 else "") +
              f'" result="b"/><feOffset in="b" dx="{num(r, -o_hi, o_hi)}" dy="{num(r, -o_hi, o_hi)}" result="o"/>'
              '<feMerge><feMergeNode in="o"/><feMergeNode in="SourceGraphic"/></feMerge></filter>')
    ops = []
    for _ in range(r.randrange(1, 4)):
        k = r.randrange(6)
        if k == 0:
            ops.append(f'<feColorMatrix type="saturate" values="{num(r, 0, 1)}"/>')
        elif k == 1:
            ops.append(f'<feColorMatrix type="hueRotate" values="{num(r, 0, 360)}"/>')
        elif k == 2:
            ops.append('<feColorMatrix type="luminanceToAlpha"/>')
        elif k == 3:
            ops.append(f'<feMorphology operator="{r.choice(["erode", "dilate"])}" radius="{num(r, 0.5, m_hi)}"/>')
        elif k == 4:
            ops.append(f'<feComposite in2="SourceGraphic" operator="{r.choice(["over", "in", "out", "atop", "xor"])}"/>')
        else:
            ops.append(f'<feComposite in2="SourceAlpha" operator="arithmetic" k1="{num(r, 0, 1)}" k2="{num(r, 0, 1)}" k3="{num(r, 0, 1)}" k4="{num(r, 0, 0.2)}"/>')
    return shadow + '<filter id="f1">' + "".join(ops) + "</filter>"


def pattern(r):
    kind = r.randrange(4)
    extra = f' patternTransform="rotate({num(r, -60, 60)}) scale({num(r, 0.6, 1.6)})"' if r.random() < 0.3 else ""
    if kind == 0:
        return f'<pattern id="p0" width="{num(r, 4, 30)}" height="{num(r, 4, 30)}" patternUnits="userSpaceOnUse"{extra}>{shape(r)}</pattern>'
    if kind == 1:
        return (f'<pattern id="p0" x="{num(r, 0, 9)}" y="{num(r, 0, 9)}" width="{num(r, 6, 30)}" height="{num(r, 6, 30)}" patternUnits="userSpaceOnUse"'
                f' viewBox="0 0 {num(r, 4, 40)} {num(r, 4, 40)}"{extra}><rect width="{num(r, 1, 20)}" height="{num(r, 1, 20)}" fill="{r.choice(COLORS[:9])}"/>'
                f'<circle cx="{num(r, 0, 20)}" cy="{num(r, 0, 20)}" r="{num(r, 1, 9)}" fill="{r.choice(COLORS[:9])}"/></pattern>')
    if kind == 2:
        return (f'<pattern id="p0" width="{num(r, 0.1, 0.6)}" height="{num(r, 0.1, 0.6)}"{extra}><rect width="{num(r, 2, 12)}" height="{num(r, 2, 12)}" fill="{r.choice(COLORS[:9])}"/>'
                f'<circle cx="{num(r, 0, 14)}" cy="{num(r, 0, 14)}" r="{num(r, 1, 6)}" fill="{r.choice(COLORS[:9])}"/></pattern>')
    return (f'<pattern id="p0" width="{num(r, 0.2, 0.6)}" height="{num(r, 0.2, 0.6)}" patternContentUnits="objectBoundingBox">'
            f'<rect width="{num(r, 0.05, 0.4)}" height="{num(r, 0.05, 0.4)}" fill="{r.choice(COLORS[:9])}"/></pattern>')


def document(r):
    units = r.choice(["", ' gradientUnits="userSpaceOnUse"'])
    lin_units = r.random() < 0.3
    lin_geo = (f' gradientUnits="userSpaceOnUse" x1="{num(r, 0, 90)}" y1="{num(r, 0, 90)}" x2="{num(r, 0, 90)}" y2="{num(r, 0, 90)}"' if lin_units else
               f' x1="{num(r, 0, 1)}" y1="{num(r, 0, 1)}" x2="{num(r, 0, 1)}" y2="{num(r, 0, 1)}"')
    lin_tr = r.choice(["", "", f' gradientTransform="skewX({num(r, -30, 30)}) scale({num(r, 0.5, 1.5)})"', f' gradientTransform="rotate({num(r, 0, 90)} 0.5 0.5)"'])
    rad_extra = r.choice(SPREADS) + r.choice(["", "", ' color-interpolation="linearRGB"', f' fr="{num(r, 0, 4)}"'])
    mask_units = r.choice(["", "", ' maskContentUnits="objectBoundingBox"'])
    mask_body = (f'<rect x="{num(r, 0, 0.5)}" y="{num(r, 0, 0.5)}" width="{num(r, 0.2, 0.9)}" height="{num(r, 0.2, 0.9)}" fill="{r.choice(["white", "#888", "red"])}"/>'
                 if mask_units else shape(r) + shape(r))
    defs = (f'<linearGradient id="g0"{lin_geo}{lin_tr}'
            f'{r.choice(SPREADS)}>{stops(r)}</linearGradient>'
            f'<radialGradient id="g1"{units} cx="{num(r, 0, 90)}" cy="{num(r, 0, 90)}" r="{num(r, 5, 60)}"'
            + (f' fx="{num(r, 0, 90)}" fy="{num(r, 0, 90)}"' if r.random() < 0.5 else "")
            + (f' gradientTransform="rotate({num(r, 0, 90)})"' if r.random() < 0.4 else "") + rad_extra + f">{stops(r)}</radialGradient>"
            f'<clipPath id="c0"{r.choice(CLIP_RULES)}>{shape(r)}</clipPath>'
            f'<clipPath id="c1" clipPathUnits="objectBoundingBox"><rect x="0.1" y="0.2" width="0.7" height="0.6"/></clipPath>'
            f'<mask id="m0"{mask_units}>{mask_body}</mask>'
            + pattern(r) +
            f'<g id="sym">{shape(r)}</g>' + filters(r) + FONT)
    head = r.choice(['width="120" height="90"', 'viewBox="0 0 150 100"', 'width="3cm" height="20mm" viewBox="-5 -5 130 95"', 'width="200" height="100" viewBox="0 0 100 50"'])
    return f'<svg xmlns="http://www.w3.org/2000/svg" {head}><defs>{defs}</defs>' + "".join(group(r, 0) for _ in range(r.randrange(1, 4))) + "</svg>"


def main() -> int:
    from svgrasterize_amd import scenedump, svg

    n = int(sys.argv[1]) if len(sys.argv) > 1 else 300
    first = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    ref = ref_loader.load()
    bad = same_error = slow = 0
    for seed in range(first, first + n):
        r = random.Random(seed)
        text = document(r)
        width = r.choice([None, None, 77, 300])
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            try:
                with time_limit(20):
                    want, _, want_size = ref.svg_scene_from_str(text, width=width, fonts=ref.FontsDB())
                ref_exc = None
            except TooSlow:
                slow += 1
                continue
            except Exception as e:  # noqa: BLE001
                ref_exc = e
            try:
                got, _, got_size = svg.svg_scene_from_str(text, width=width)
                my_exc = None
            except Exception as e:  # noqa: BLE001
                my_exc = e
        if ref_exc is not None or my_exc is not None:
            if (ref_exc is None) != (my_exc is None):
                bad += 1
                print(f"seed {seed}: reference {'raised ' + repr(ref_exc) if ref_exc else 'loaded'}, here {'raised ' + repr(my_exc) if my_exc else 'loaded'}")
            else:
                same_error += 1
            continue
        if (want is None) != (got is None):
            bad += 1
            print(f"seed {seed}: empty scene on one side only")
            continue
        if want is None:
            continue
        d = gen_golden.Dumper(ref)
        try:
            with time_limit(20):
                tree_ref = json.loads(json.dumps(d.node(want)))
        except TooSlow:  # (the reference's stroker spinning on an outline; the native one here finishes)
            slow += 1
            continue
        except Exception:  # noqa: BLE001  (e.g. a stroke the reference's own stroker cannot outline)
            same_error += 1
            continue
        try:
            tree, arrays = scenedump.dump_scene(got)
        except Exception as e:  # noqa: BLE001
            bad += 1
            print(f"seed {seed}: the reference outlined every stroke, here: {e!r}")
            continue
        diffs = scenedump.compare_dumps(tree, arrays, tree_ref, d.arrays(), 1e-11)
        if [float(v) for v in want_size] != [float(v) for v in got_size]:
            diffs.append(f"size {want_size} vs {got_size}")
        if diffs:
            bad += 1
            print(f"seed {seed}: " + "; ".join(diffs[:3]))
    print(f"{n} documents, {bad} mismatches, {same_error} rejected by both, {slow} given up on (reference > 20 s)")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
