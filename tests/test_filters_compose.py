"""the compose modes, filter primitives and the luminance mask the reference implements besides the
hot path -- Layer.compose OUT / ATOP / XOR / arithmetic, Layer.color_matrix, Layer.morphology, Filter chains with
feOffset / feMerge / feBlend / feComposite / feColorMatrix / feMorphology, Scene MASK -- against results produced by the
reference itself (tests/golden/filter_kat.npz, oracle/gen_golden.py --only filters)."""
import json
import os
import warnings

import numpy as np
import pytest

from tests.util import assert_close64

GOLD = os.path.join(os.path.dirname(__file__), "golden")
pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def kat():
    z = np.load(os.path.join(GOLD, "filter_kat.npz"))
    return z, json.loads(str(z["meta"]))


@pytest.fixture(scope="module")
def S():
    import svgrasterize_amd as S

    S.Context.get()
    return S


def test_compose_modes(S, kat):
    z, meta = kat
    for k, m in enumerate(meta["compose"]):
        layers = [S.Layer(z[f"c{k}_in{j}"], tuple(l["offset"]), l["pre_alpha"], l["linear_rgb"]) for j, l in enumerate(m["layers"])]
        mode = tuple(m["mode"]) if isinstance(m["mode"], list) else m["mode"]
        out = S.Layer.compose(layers, mode, linear_rgb=m["linear_rgb"])
        assert [int(v) for v in out.offset] == m["out_offset"] and (out.pre_alpha, out.linear_rgb) == (m["out_pre_alpha"], m["out_linear_rgb"])
        assert_close64(out.image, z[f"c{k}_out"], atol=1e-14, what=f"compose case {k} mode {mode}")
    with pytest.raises(ValueError):
        S.Layer.compose(layers, 17)
    with pytest.raises(ValueError):
        S.Layer.compose(layers, (1.0, 2.0))


def test_color_matrix_and_morphology(S, kat):
    z, meta = kat
    for j, m in enumerate(meta["cmatrix"]):
        layer = S.Layer(z[f"m{j}_in"], tuple(m["offset"]), m["pre_alpha"], m["linear_rgb"])
        out = layer.color_matrix(z[f"m{j}_matrix"])
        assert (out.pre_alpha, out.linear_rgb) == (m["out_pre_alpha"], m["out_linear_rgb"]) and tuple(out.offset) == tuple(m["offset"])
        assert_close64(out.image, z[f"m{j}_out"], atol=1e-14, what=f"color matrix {j}")
        assert np.array_equal(layer.image, z[f"m{j}_in"])  # the input layer is untouched
    with pytest.raises(ValueError):
        layer.color_matrix(np.eye(4))
    for j, m in enumerate(meta["morph"]):
        layer = S.Layer(z[f"p{j}_in"], tuple(m["offset"]), m["pre_alpha"], m["linear_rgb"])
        out = layer.morphology(m["x"], m["y"], m["method"])
        assert [int(v) for v in out.offset] == m["out_offset"] and out.image.shape == z[f"p{j}_out"].shape
        assert_close64(out.image, z[f"p{j}_out"], atol=1e-14, what=f"morphology {j}")
    with pytest.raises(ValueError):
        layer.morphology(2, 2, "mean-ish")
    with pytest.raises(ValueError):
        layer.morphology(500, 2, "max")


def test_filter_chains(S, kat):
    z, meta = kat
    for j, m in enumerate(meta["chain"]):
        flt = S.Filter.empty()
        for i, prim in enumerate(m["prims"]):
            attrs = []
            for a in prim["attrs"]:
                if a == "matrix":
                    attrs.append(z[f"f{j}_matrix{i}"])
                elif isinstance(a, list):
                    attrs.append(tuple(a))
                else:
                    attrs.append(a)
            flt = S.Filter(flt.names, flt.filters + [(prim["type"], tuple(attrs), prim["inputs"])])
        src = S.Layer(z[f"f{j}_in"], tuple(m["in_offset"]), True, False)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            out = flt(S.Transform(z[f"f{j}_tr"]), src)
        assert [int(v) for v in out.offset] == m["out_offset"], m["name"]
        assert (out.pre_alpha, out.linear_rgb) == (m["out_pre_alpha"], m["out_linear_rgb"]), m["name"]
        assert_close64(out.image, z[f"f{j}_out"], atol=1e-13, what=f"filter chain {m['name']}")
    # the builder methods produce the reference's (type, attrs, inputs) triples
    f = S.Filter.empty().blur(1.0, result="b").offset(1, 2, input="b").merge(["b", S.FE_SOURCE_GRAPHIC]).composite("b", None, 3)
    assert [t for t, _a, _i in f.filters] == [S.FE_GAUSSIAN_BLUR, S.FE_OFFSET, S.FE_MERGE, S.FE_COMPOSITE]
    assert f.filters[1][2] == [2] and f.filters[2][2] == [2, 1] and f.filters[3][2] == [2, 4]


def test_luminance_mask_scene(S, kat):
    z, meta = kat
    swap = S.Transform().matrix(0, 1, 0, 1, 0, 0)
    blob = S.Path.from_svg("M20,30 C20,5 80,5 80,30 S110,85 60,90 C30,92 20,60 20,30 Z")
    ring = S.Path.from_svg("M60,10 L75,95 L10,40 L110,40 L45,95 Z")
    for j, m in enumerate(meta["mask"]):
        target = S.Scene.fill(blob, np.array([0.2, 0.5, 0.1, 0.9]))
        mask_scene = S.Scene.group([S.Scene.fill(ring, np.array([0.9, 0.9, 0.2, 1.0]), "evenodd"),
                                    S.Scene.fill(blob, np.array([0.1, 0.3, 0.6, 0.7]))])
        layer, _hull = target.mask(mask_scene, False).render(swap, viewport=[0, 0, 120, 140], linear_rgb=m["linear_rgb"])
        assert [int(v) for v in layer.offset] == m["offset"] and (layer.pre_alpha, layer.linear_rgb) == (m["pre_alpha"], m["out_linear_rgb"])
        assert_close64(layer.image, z[f"k{j}_out"], atol=1e-12, what=f"luminance mask {j}")


def test_module_level_canvas_functions(S):
    """canvas_create / canvas_compose / canvas_merge_at / canvas_merge_union / canvas_merge_intersect (S:235-416) called
    directly on arrays, against the reference's own results (tests/golden/canvasfn_kat.npz, gen_golden.py --only canvasfn)."""
    from functools import partial

    z = np.load(os.path.join(GOLD, "canvasfn_kat.npz"))
    meta = json.loads(str(z["meta"]))
    for i, m in enumerate(meta):
        fn = m["fn"]
        mode = tuple(m["mode"]) if isinstance(m.get("mode"), list) else m.get("mode")
        if fn == "compose":
            got = S.canvas_compose(mode, z[f"{i}_dst"], z[f"{i}_src"])
            assert_close64(got, z[f"{i}_out"], atol=1e-15, what=f"canvas_compose case {i} mode {mode}")
        elif fn == "merge_at":
            base = z[f"{i}_base"].copy()
            res = S.canvas_merge_at(base, z[f"{i}_over"], tuple(m["offset"]))
            assert (res is None) == m["none"] and (res is None or res is base)
            assert_close64(base, z[f"{i}_out"], atol=1e-15, what=f"canvas_merge_at case {i}")
        elif fn in ("union", "intersect"):
            layers = [(z[f"{i}_in{j}"], tuple(o)) for j, o in enumerate(m["offsets"])]
            blend = partial(S.canvas_compose, mode)
            if fn == "union":
                img, off = S.canvas_merge_union(layers, full=m["full"], blend=blend)
            else:
                img, off = S.canvas_merge_intersect(layers, blend=blend)
            assert [int(v) for v in off] == m["offset"]
            assert_close64(img, z[f"{i}_out"], atol=1e-15, what=f"canvas_merge_{fn} case {i}")
        elif fn == "intersect_none":
            assert S.canvas_merge_intersect([(np.ones((3, 3, 4)), (0, 0)), (np.ones((3, 3, 4)), (10, 10))]) is None
    canvas, tr = S.canvas_create(5, 3, bg=np.array([0.1, 0.2, 0.3, 1.0]))
    assert np.array_equal(canvas, z["create_canvas"]) and np.array_equal(np.asarray(tr.m, dtype=np.float64), z["create_m"])
    one = [(np.ones((2, 2, 4)), (1, 1))]
    assert S.canvas_merge_union(one)[0] is one[0][0] and S.canvas_merge_intersect(one)[0] is one[0][0]
    for f in (S.canvas_merge_union, S.canvas_merge_intersect):
        with pytest.raises(ValueError):
            f([])
    with pytest.raises(ValueError):
        S.canvas_compose(17, np.zeros((2, 2, 4)), np.zeros((2, 2, 4)))
    # a caller's own blend function is honoured as is
    mine = lambda d, s: d * 0.5 + s * 0.25  # noqa: E731
    img, off = S.canvas_merge_union([(np.ones((2, 2, 4)), (0, 0)), (np.ones((2, 2, 4)), (1, 1))], blend=mine)
    assert off == (0, 0) and img[1, 1, 0] == 0.75 and img[0, 0, 0] == 0.5 and img[2, 2, 0] == 0.25
