"""Pin the CPU oracle (oracle/svgr_oracle.c) against fixtures produced by the reference itself
(oracle/gen_golden.py).  Runs without a GPU."""
import numpy as np
import pytest

from oracle import oracle as orc
from tests.util import load, meta, assert_close64


def test_line_coverage_kat_bit_exact():
    g = load("coverage_kat.npz")
    for i in range(len(g["h"])):
        h, w = int(g["h"][i]), int(g["w"][i])
        ref = g["trace"][g["trace_off"][i]: g["trace_off"][i + 1]].reshape(h, w)
        got = orc.line_coverage(np.zeros((h, w)), g["lines"][i])
        assert np.array_equal(got, ref), f"case {i}: line {g['lines'][i].tolist()} on {h}x{w}"


def test_survey_kat_values():
    # SURVEY 8c-1 printed values (6 d.p.)
    t = orc.line_coverage(np.zeros((4, 6)), [[0.5, 0.25], [3.5, 5.75]])
    assert np.allclose(t[0], [0.153409, 0.339015, 0.007576, 0, 0, 0], atol=1e-6)
    assert np.allclose(t.sum(axis=1), [0.5, 1, 1, 0.346591], atol=1e-6)
    r = orc.line_coverage(np.zeros((4, 6)), [[3.5, 5.75], [0.5, 0.25]])
    assert np.array_equal(r, -t)
    left = orc.line_coverage(np.zeros((3, 4)), [[-1, -2.5], [2, -0.5]])
    assert np.array_equal(left[:, 0], [1.0, 1.0, 0.0]) and not left[:, 1:].any()


@pytest.mark.parametrize("name", ["rand_small", "rand_big", "tiny_curves", "degenerate", "tiger512"])
def test_flatten_bit_exact(name):
    g = load("flatten_kat.npz")
    batch = g[name + "_in"]
    assert np.array_equal(orc.flatness(batch), g[name + "_flatness"])
    assert np.array_equal(orc.split(batch), g[name + "_split"])
    assert np.array_equal(orc.flatten(batch, 0.1), g[name + "_edges"])  # same order as the reference


def test_transform_bit_exact():
    g = load("flatten_kat.npz")
    for m, pts, out in zip(g["tr_m"], g["tr_in"], g["tr_out"]):
        assert np.array_equal(orc.transform_points(m, pts), out)


def test_mask_kat():
    g = load("mask_kat.npz")
    for idx, m in enumerate(meta(g)):
        res = orc.path_mask(g[f"{idx}_lines"], g[f"{idx}_cubics"], g[f"{idx}_tr"], m["rule"], m["viewport"])
        if m["none"]:
            assert res is None, m["name"]
            continue
        assert res is not None, m["name"]
        mask, off, _edges = res
        assert list(off) == m["offset"], m["name"]
        ref = g[f"{idx}_image"]
        assert mask.shape == ref.shape[:2], m["name"]
        if m["has_paint"]:
            paint = orc.paint_for_fill(g[f"{idx}_paint"], m["linear_rgb"])
            img = orc.fill_solid(mask, paint)
            # libm pow vs numpy pow on the 4-vector paint may differ in the last f64 bit
            assert_close64(img, ref, atol=2e-16, what=m["name"])
        else:
            assert np.array_equal(mask, ref[..., 0]), m["name"]


def test_compose_kat():
    g = load("compose_kat.npz")
    for idx, m in enumerate(meta(g)):
        ins = [(g[f"{idx}_in{j}"], tuple(d["offset"])) for j, d in enumerate(m["in"])]
        if m["tag"] in ("over", "over_convert"):
            conv = []
            for (im, off), d in zip(ins, m["in"]):
                if im.shape[2] == 4:
                    im = orc.convert(im, d["pre_alpha"], d["linear_rgb"], True, m["linear_rgb"])
                conv.append((im, off))
            out, off = orc.compose_over(conv)
            assert list(off) == m["offset"]
            tol = 0.0 if m["tag"] == "over" else 4e-16
            assert_close64(out, g[f"{idx}_out"], atol=tol, what=f"{m['tag']} {idx}")
        elif m["tag"] in ("in", "in_empty"):
            res = orc.compose_in(ins)
            if m["none"]:
                assert res is None
            else:
                out, off = res
                assert list(off) == m["offset"]
                assert np.array_equal(out, g[f"{idx}_out"])
        elif m["tag"] == "convert":
            d = m["in"][0]
            out = orc.convert(ins[0][0], d["pre_alpha"], d["linear_rgb"], m["to_pre_alpha"], m["to_linear_rgb"])
            assert_close64(out, g[f"{idx}_out"], atol=4e-16, what=f"convert {idx}")


# ------------------------------------------------------------------------------------------
# the whole CPU render (the checker of the bench line) against the reference's own render
# ------------------------------------------------------------------------------------------
@pytest.mark.parametrize("key", ["s256_n48", "s700_n300"])
def test_render_solid_equals_the_reference_on_the_bench_generator(key):
    """svgrasterize.py_amd/synth.py scenes drawn by the reference (Path.fill + Layer.compose(OVER) + canvas_merge_at, fixture made by
    oracle/gen_golden.py --only synth) against oracle.render_solid: what bench.py checks the timed canvas with."""
    import json

    from svgrasterize_amd import synth

    g = load("synth_kat.npz")
    m = next(x for x in json.loads(str(g["meta"])) if x["key"] == key)
    sc = synth.make_scene(m["size"], m["paths"])
    got, P, E = orc.render_solid(synth.presentation_segs(sc), sc["seg_kind"], sc["path_seg_off"], sc["path_rule"], sc["path_paint"],
                                 sc["viewport"], clip01=True)
    ref = g[key + "_canvas"]
    assert got.shape == ref.shape
    err = np.abs(got - ref)
    # (sequential OVER per pixel in both; the reference's mask * paint and 1 - src_a are the oracle's: rounding-level agreement)
    assert err.max() <= 4e-16, f"{key}: max |oracle - reference| = {err.max():.3e}"
    assert P > 0 and E > 0


def test_render_solid_on_the_tiger_leaves_equals_the_reference_canvas():
    """Ghostscript tiger at 1/16 scale: the scene's paint-ordered leaves (fills and stroked outlines, transforms applied on the host)
    through oracle.render_solid against the canvas the reference drew (scene_tiger.npz::s128_canvas)."""
    import os

    import svgrasterize_amd as S
    from svgrasterize_amd import scenedump

    scene, info, z = scenedump.load_scene(os.path.join(os.path.dirname(__file__), "golden", "scene_tiger.npz"))
    r = next(r for r in info["renders"] if r["tag"] == "s128")
    tr = S.Transform().matrix(0, 1, 0, 1, 0, 0).scale(r["scale"])
    hh, ww = r["size"]
    leaves = scene.leaves(tr, linear_rgb=False)
    assert leaves, "the tiger is solid fills and strokes: one batchable run"
    segs, kinds, offs, rules, paints = [], [], [0], [], []
    for leaf in leaves:
        path, m6, rule, paint, flags = leaf[:5]
        assert flags == 0
        s, k = path.packed()
        m = np.eye(3)
        m[:2, :] = np.asarray(m6, dtype=np.float64).reshape(2, 3)
        segs.append(orc.transform_points(m, s.reshape(-1, 4, 2)).reshape(-1, 8))
        kinds.append(k)
        offs.append(offs[-1] + len(s))
        rules.append(rule)
        paints.append(paint)
    got, P, E = orc.render_solid(np.concatenate(segs), np.concatenate(kinds), np.array(offs, dtype=np.int64), np.array(rules, dtype=np.uint8),
                                 np.array(paints), (0, 0, hh, ww), clip01=True)
    ref = z["s128_canvas"]
    assert got.shape == ref.shape
    # (the reference composites group by group, the batch path by path: source-over is associative up to double rounding)
    assert np.abs(got - ref).max() <= 1e-12
