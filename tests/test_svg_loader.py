"""SVG front-end (svgrasterize.py_amd/svg.py): the Scene it builds against the Scene the reference's
own loader built from the same document -- as scene dumps (tests/golden/scene_*.npz for the reference's demo files, read
from /root/reference/demo when that exists: the build container only; tests/golden/svg_kat.npz for the hand-written
documents of tests/svg_cases.py, everywhere)."""
import json
import os
import warnings

import numpy as np
import pytest

from tests import svg_cases

GOLD = os.path.join(os.path.dirname(__file__), "golden")
DEMO = "/root/reference/demo"


def _compare(scene, tree_ref, arrays_ref, tol=1e-12):
    from svgrasterize_amd import scenedump

    tree, arrays = scenedump.dump_scene(scene)
    return scenedump.compare_dumps(tree, arrays, tree_ref, arrays_ref, tol)


def test_handwritten_documents_match_reference_loader():
    from svgrasterize_amd import svg

    z = np.load(os.path.join(GOLD, "svg_kat.npz"))
    meta = json.loads(str(z["meta"]))
    assert [m["name"] for m in meta] == [name for name, _text, _w in svg_cases.CASES]
    for idx, ((name, text, width), m) in enumerate(zip(svg_cases.CASES, meta)):
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            scene, _ids, size = svg.svg_scene_from_str(text, width=width)
        if m["none"]:
            assert scene is None, name
            continue
        assert size is None and m["size"] is None or [float(v) for v in size] == m["size"], name
        arrays = {k: z[f"{idx}_{k}"] for k in ("lines", "cubics", "line_off", "cubic_off")}
        diffs = _compare(scene, json.loads(str(z[f"{idx}_tree"])), arrays)
        assert not diffs, f"{name}: " + "; ".join(diffs[:5])


@pytest.mark.parametrize("name,svg_file,width", [("tiger", "icons/tiger.svg", 2048), ("material", "material-design.svg", 4096),
                                                  ("icons", "icons.svg", None), ("prompt", "prompt.svg", 256)])
def test_demo_documents_match_reference_dumps(name, svg_file, width):
    path = os.path.join(DEMO, svg_file)
    if not os.path.exists(path):
        pytest.skip("the reference's demo files exist only in the build container")
    from svgrasterize_amd import svg

    z = np.load(os.path.join(GOLD, f"scene_{name}.npz"))
    info = json.loads(str(z["info"]))
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        fonts = svg.FontsDB()
        fonts.register_file(os.path.join(os.path.dirname(DEMO), "fonts.svgz"))  # prompt.svg is all text
        scene, _ids, size = svg.svg_scene_from_filepath(path, width=width, fonts=fonts)
    w, h = size
    assert [int(h), int(w)] == info["size"]
    arrays = {k: z[k] for k in ("lines", "cubics", "line_off", "cubic_off")}
    diffs = _compare(scene, json.loads(str(z["tree"])), arrays)
    assert not diffs, "; ".join(diffs[:8])


def test_scalar_parsers():
    from svgrasterize_amd import svg

    assert svg.parse_size("1in") == 96 and svg.parse_size("2.54cm") == pytest.approx(96) and svg.parse_size("12pt") == 16
    assert svg.parse_size("3") == 3.0 and svg.parse_size(None, 7) == 7 and svg.parse_size("2em") == 24
    assert svg.parse_float("50%") == 0.5 and svg.parse_float("3px") == 3.0 and svg.parse_float(None) is None
    assert svg.parse_floats("1,2  3") == [1.0, 2.0, 3.0]
    with pytest.raises(ValueError):
        svg.parse_floats("1 2", 4, 4)
    t = svg.parse_transform("translate(10, 5) scale(2) rotate(90)")
    assert np.allclose(t([[1, 0]]), [[10, 7]])
    with pytest.raises(ValueError):
        svg.parse_transform("spin(3)")
    assert np.allclose(svg.parse_color("#fff"), [1, 1, 1, 1]) and np.allclose(svg.parse_color("red"), [1, 0, 0, 1])
    c = svg.parse_color("rgba(255, 0, 0, 127.5)")
    assert np.allclose(c, [0.5, 0, 0, 0.5])
    assert svg.parse_paint("none", {}) is None and svg.parse_paint(None, {}) is None


def test_fonts_db_resolution_and_glyph_layout():
    """fonts.py on its own: family fallbacks, style / weight choice, ligatures, kerning, the missing glyph."""
    from svgrasterize_amd.fonts import Font, FontsDB, Glyph

    def face(family, weight, style="normal"):
        f = Font(family, weight, style, 800, -200, 1000)
        f.glyphs.update({c: Glyph(c, 500.0, "M0,0 H100 V100 z", c) for c in "abf"})
        f.glyphs["ff"] = Glyph("ff", 800.0, "M0,0 H200 V100 z", "ff")
        f.missing_glyph = Glyph(None, 300.0, "", "missing-glyph")
        f.hkern[("a", "b")] = 50.0
        return f

    db = FontsDB()
    assert db.resolve("anything") is None
    regular, bold, italic, mono = face("Serif", 400), face("Serif", 700), face("Serif", 400, "italic"), face("monospace", 400)
    for f in (regular, bold, italic, mono):
        db.register(f)
    db.register(bold, alias="heavy")
    assert db.resolve(None) is regular and db.resolve("Georgia") is regular and db.resolve("no idea") is regular
    assert db.resolve("serif", 650) is bold and db.resolve("heavy") is bold
    assert db.resolve("serif", None, "italic") is italic and db.resolve("serif", None, "oblique") is regular
    assert db.resolve("Courier") is mono and db.resolve("Some Mono") is mono
    assert db.resolve("Arial") is None  # no sans family registered, and the fallback of the fallback is absent

    placed, advance = regular.str_to_glyphs("abffa?")
    assert [(x, g.name) for x, g in placed] == [(0.0, "a"), (450.0, "b"), (950.0, "ff"), (1750.0, "a"), (2250.0, "missing-glyph")]
    assert advance == 2550.0
    path, width = regular.str_to_path(20, "ab")
    assert width == 950.0 * (20 / 1000) and len(path.subpaths) == 2
    assert np.allclose(np.asarray(path.subpaths[1][0][1]), [[9.0, -0.0], [11.0, -0.0]])


@pytest.mark.gpu
def test_documents_render_like_the_reference():
    """Whole front-end + hot path: document text -> Scene -> HIP render, against the canvas the reference drew."""
    import svgrasterize_amd as S
    from svgrasterize_amd import svg
    from tests.util import assert_f32_1ulp

    S.Context.get()
    z = np.load(os.path.join(GOLD, "svg_kat.npz"))
    meta = json.loads(str(z["meta"]))
    drawn = drawn_lin = 0
    for idx, ((name, text, width), m) in enumerate(zip(svg_cases.CASES, meta)):
        if not m.get("canvas"):
            continue
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            scene, _ids, _size = svg.svg_scene_from_str(text, width=width)
        h, w = m["canvas"]
        layer, _hull = scene.render(S.Transform().matrix(0, 1, 0, 1, 0, 0), viewport=[0, 0, h, w], linear_rgb=False)
        assert_f32_1ulp(layer.convert(pre_alpha=True, linear_rgb=False).to_canvas_f32(h, w), z[f"{idx}_canvas"], what=f"{name} canvas")
        drawn += 1
        if f"{idx}_canvas_lin" in z.files:  # the same document composited in linear RGB (--linear-rgb)
            layer, _hull = scene.render(S.Transform().matrix(0, 1, 0, 1, 0, 0), viewport=[0, 0, h, w], linear_rgb=True)
            assert layer.linear_rgb
            lin = layer.convert(pre_alpha=True, linear_rgb=True).to_canvas_f32(h, w)
            assert_f32_1ulp(lin, z[f"{idx}_canvas_lin"], what=f"{name} canvas, linear RGB")
            drawn_lin += 1
    assert drawn >= 8 and drawn_lin >= 4


def _png_pixels(png: bytes) -> np.ndarray:
    import struct
    import zlib

    pos, idat, shape = 8, b"", None
    while pos < len(png):
        n, tag = struct.unpack(">I4s", png[pos:pos + 8])
        data = png[pos + 8: pos + 8 + n]
        if tag == b"IHDR":
            w, h = struct.unpack(">II", data[:8])
            shape = (h, w)
        elif tag == b"IDAT":
            idat += data
        pos += 12 + n
    raw = np.frombuffer(zlib.decompress(idat), dtype=np.uint8).reshape(shape[0], shape[1] * 4 + 1)
    assert not raw[:, 0].any()  # filter type 0 on every scanline, like the reference's writer
    return raw[:, 1:].reshape(shape[0], shape[1], 4)


@pytest.mark.gpu
def test_render_svg_writes_the_reference_png():
    """Document text -> PNG in one call, against the file the reference's command line writes.  Same size, same chunks;
    a channel may differ by one level only where the value in front of ``round(x * 255)`` sits on a rounding tie
    (0.5 * 255 = 127.5: coverage 1 - 1e-16 vs 1 decides it, and double sums in another order differ by that much);
    documents without such ties are byte-identical."""
    import io

    import svgrasterize_amd as S
    from svgrasterize_amd import svg

    S.Context.get()
    z = np.load(os.path.join(GOLD, "svg_kat.npz"))
    meta = json.loads(str(z["meta"]))
    bg = svg.parse_color("#fdf6e3")
    checked, identical = 0, 0
    for idx, ((name, text, width), m) in enumerate(zip(svg_cases.CASES, meta)):
        if not m.get("canvas"):
            continue
        h, w = m["canvas"]
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            scene, _ids, _size = svg.svg_scene_from_str(text, width=width)
            layer, _hull = scene.render(S.Transform().matrix(0, 1, 0, 1, 0, 0), viewport=[0, 0, h, w], linear_rgb=False)
            page = layer.convert(pre_alpha=True, linear_rgb=False).on_canvas(h, w)
            for key, colour in (("png", None), ("png_bg", bg)):
                png = svg.render_svg(io.StringIO(text), width=width, bg=colour)
                want = z[f"{idx}_{key}"].tobytes()
                if png == want:
                    identical += 1
                    continue
                got_px, want_px = _png_pixels(png), _png_pixels(want)
                assert got_px.shape == want_px.shape == (h, w, 4), name
                final = page if colour is None else page.background(colour)
                levels = final.convert(pre_alpha=False, linear_rgb=False).image * 255.0
                differs = got_px != want_px
                assert np.abs(got_px.astype(int) - want_px.astype(int)).max() <= 1, name
                assert (np.abs(levels[differs] - np.floor(levels[differs]) - 0.5) < 1e-6).all(), f"{name}: not a rounding tie"
                assert differs.any(axis=-1).mean() < 0.05, name
        checked += 1
    assert checked >= 8 and identical >= 6
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        assert svg.render_svg(io.StringIO(svg_cases.CASES[-1][1])) is None  # the empty document
    # one element by id, on its own bounding box, with an extra transform and a default foreground colour
    text = dict((n, t) for n, t, _w in svg_cases.CASES)["nested_svg_use"].replace(' fill="green"', "")
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        png = svg.render_svg(io.StringIO(text), width=400, id="leaf", fg=svg.parse_color("#b5651d"),
                             transform=svg.parse_transform("scale(3) rotate(10)"))
    want = z["by_id_png"].tobytes()
    if png != want:
        got_px, want_px = _png_pixels(png), _png_pixels(want)
        assert got_px.shape == want_px.shape
        assert np.abs(got_px.astype(int) - want_px.astype(int)).max() <= 1
        assert (got_px != want_px).any(axis=-1).mean() < 0.02
    with pytest.raises(KeyError):
        svg.render_svg(io.StringIO(text), id="no-such-element")


@pytest.mark.gpu
def test_generated_documents_render_like_the_reference():
    """40 grammar-generated documents (random nestings of groups, viewports, clips, masks, patterns, gradients, strokes;
    tests/golden/svg_fuzz_kat.npz stores the text and the canvas the reference drew): loader + hot path end to end."""
    import svgrasterize_amd as S
    from svgrasterize_amd import svg
    from tests.util import assert_f32_1ulp

    S.Context.get()
    z = np.load(os.path.join(GOLD, "svg_fuzz_kat.npz"))
    meta = json.loads(str(z["meta"]))
    assert len(meta) >= 36
    for k, m in enumerate(meta):
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            scene, _ids, size = svg.svg_scene_from_str(m["text"], width=m["width"])
            h, w = m["size"]
            assert [int(size[1]), int(size[0])] == [h, w]
            layer, _hull = scene.render(S.Transform().matrix(0, 1, 0, 1, 0, 0), viewport=[0, 0, h, w], linear_rgb=False)
        assert_f32_1ulp(layer.convert(pre_alpha=True, linear_rgb=False).to_canvas_f32(h, w), z[f"{k}_canvas"].astype(np.float64), what=f"generated document {m['seed']}")
        if m.get("crop"):  # the same document through a window somewhere on (or partly off) the page
            r0, c0, rows, cols = m["crop"]
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                res = scene.render(S.Transform().matrix(0, 1, 0, 1, 0, 0), viewport=[r0, c0, rows, cols], linear_rgb=False)
            win = np.zeros((rows, cols, 4), dtype=np.float32)
            if res is not None:
                win = res[0].convert(pre_alpha=True, linear_rgb=False).translate(-r0, -c0).to_canvas_f32(rows, cols)
            assert_f32_1ulp(win, z[f"{k}_canvas_crop"].astype(np.float64), what=f"generated document {m['seed']}, window {m['crop']}")
        if m.get("free"):  # ... and without a viewport: the layer grows to whatever the content reaches
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                res = scene.render(S.Transform().matrix(0, 1, 0, 1, 0, 0), linear_rgb=False)
            free = res[0].convert(pre_alpha=True, linear_rgb=False)
            assert [int(free.offset[0]), int(free.offset[1])] == m["free"] and free.image.shape == z[f"{k}_layer_free"].shape
            assert_f32_1ulp(free.image.astype(np.float32), z[f"{k}_layer_free"].astype(np.float64), what=f"generated document {m['seed']}, no viewport")
        if f"{k}_canvas_lin" in z.files:
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                layer, _hull = scene.render(S.Transform().matrix(0, 1, 0, 1, 0, 0), viewport=[0, 0, h, w], linear_rgb=True)
            assert_f32_1ulp(layer.convert(pre_alpha=True, linear_rgb=True).to_canvas_f32(h, w), z[f"{k}_canvas_lin"].astype(np.float64),
                            what=f"generated document {m['seed']}, linear RGB")
