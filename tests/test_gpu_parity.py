"""GPU parity tests: the HIP path (through the C ABI) against the reference's golden fixtures and
against the CPU oracle on the same seeded inputs.  Run with `pytest -m gpu` on an MI355X."""
import json
import os

import numpy as np
import pytest

from tests.util import GOLDEN, assert_close64, assert_f32_1ulp, f32_contract_counts, load, meta, sort_edges

pytestmark = pytest.mark.gpu

F64_TOL = 1e-11  # double results differ from the reference only by summation order (atomics, tile carry-in)


@pytest.fixture(scope="module")
def S():
    import svgrasterize_amd as S

    S.Context.get()  # fails loudly without a gfx950 device / built library
    return S


@pytest.fixture(scope="module")
def orc():
    from oracle import oracle

    oracle.lib()
    return oracle


def test_context(S):
    name = S.Context.get().name()
    assert "gfx950" in name, name


# ------------------------------------------------------------------------------------------
# flatten: edge SET is bit-exact with the reference (order is curve order instead of level order)
# ------------------------------------------------------------------------------------------
@pytest.mark.parametrize("name", ["rand_small", "rand_big", "tiny_curves", "degenerate", "tiger512"])
def test_flatten_edge_set_bit_exact(S, name):
    from svgrasterize_amd import _abi

    g = load("flatten_kat.npz")
    cubics = g[name + "_in"].reshape(-1, 8)
    n = len(cubics)
    ident = np.array([1.0, 0, 0, 0, 1, 0])
    batch = _abi.Batch(S.Context.get(), cubics, np.ones(n, np.uint8), [0, n], ident, [0], [[0, 0, 0, 1]], viewport=None)
    st = batch.plan()
    edges, edge_path = batch.edges()
    ref = g[name + "_edges"]
    assert st.n_edges == len(ref)
    assert np.array_equal(sort_edges(edges), sort_edges(ref))
    assert not edge_path.any()


def test_transform_on_device_bit_exact(S):
    """Transform.__call__ fma form (S:531-534): straight lines through the device transform."""
    from svgrasterize_amd import _abi

    g = load("flatten_kat.npz")
    for m, pts, out in zip(g["tr_m"][:8], g["tr_in"][:8], g["tr_out"][:8]):
        lines_in = pts.reshape(-1, 2, 2)
        segs = np.zeros((len(lines_in), 8))
        segs[:, :4] = lines_in.reshape(-1, 4)
        batch = _abi.Batch(S.Context.get(), segs, np.zeros(len(segs), np.uint8), [0, len(segs)], m[:2].reshape(6), [0],
                           [[0, 0, 0, 1]], viewport=None)
        batch.plan()
        edges, _ = batch.edges()
        assert np.array_equal(sort_edges(edges), sort_edges(out))  # slot order depends on which wave reserves first


# ------------------------------------------------------------------------------------------
# Path.mask / Path.fill against the reference's own outputs
# ------------------------------------------------------------------------------------------
def _cases():
    g = load("mask_kat.npz")
    return g, meta(g)


def test_mask_fill_kat(S):
    g, cases = _cases()
    for idx, m in enumerate(cases):
        path = S.Path.from_segments(g[f"{idx}_segt"], g[f"{idx}_segp"], g[f"{idx}_subs"])
        tr = S.Transform(g[f"{idx}_tr"])
        if m["has_paint"]:
            res = path.fill(tr, g[f"{idx}_paint"], fill_rule=m["rule"], viewport=m["viewport"], linear_rgb=m["linear_rgb"])
        else:
            res = path.mask(tr, fill_rule=m["rule"], viewport=m["viewport"])
        if m["none"]:
            assert res is None, m["name"]
            continue
        assert res is not None, m["name"]
        layer, hull = res
        ref = g[f"{idx}_image"]
        assert [int(layer.offset[0]), int(layer.offset[1])] == m["offset"], m["name"]  # integer pixel indices: exact
        assert layer.image.shape == ref.shape, m["name"]
        assert layer.pre_alpha == m["pre_alpha"] and layer.linear_rgb == m["layer_linear_rgb"]
        assert_close64(layer.image, ref, atol=F64_TOL, what=m["name"])
        assert_f32_1ulp(layer.image.astype(np.float32), ref, what=m["name"])
        if m["viewport"] is None:  # the hull is only compared where arcs/quads are absent or host conversion is exact
            assert np.allclose(np.array(hull.points), g[f"{idx}_hull"], atol=1e-9), m["name"]


def test_mask_errors(S):
    p = S.Path.from_svg("M1,1 L5,1 L3,4 Z")
    with pytest.raises(ValueError):
        p.mask(S.Transform(), fill_rule="bogus")
    assert S.Path([]).mask(S.Transform()) is None
    assert p.fill(S.Transform(), None) is None


def test_path_data_reader_matches_fixture_segments(S):
    g, cases = _cases()
    for idx, m in enumerate(cases):
        a = S.Path.from_svg(m["d"]).packed()[0]
        b = S.Path.from_segments(g[f"{idx}_segt"], g[f"{idx}_segp"], g[f"{idx}_subs"]).packed()[0]
        assert a.shape == b.shape and np.allclose(a, b, atol=1e-9), m["name"]


# ------------------------------------------------------------------------------------------
# Layer.compose / convert / opacity
# ------------------------------------------------------------------------------------------
def test_compose_kat(S):
    g = load("compose_kat.npz")
    for idx, m in enumerate(meta(g)):
        ins = [S.Layer(g[f"{idx}_in{j}"].copy(), tuple(d["offset"]), d["pre_alpha"], d["linear_rgb"]) for j, d in enumerate(m["in"])]
        tag = m["tag"]
        if tag in ("over", "over_convert", "in", "in_empty", "full"):  # "full": OUT / ATOP / XOR on the union canvas
            out = S.Layer.compose(ins, m["method"], m["linear_rgb"])
        elif tag == "convert":
            out = ins[0].convert(pre_alpha=m["to_pre_alpha"], linear_rgb=m["to_linear_rgb"])
        elif tag == "opacity":
            out = ins[0].opacity(m["opacity"], m["linear_rgb"])
        if m["none"]:
            assert out is None
            continue
        assert list(map(int, out.offset)) == m["offset"], (tag, idx)
        assert out.pre_alpha == m["pre_alpha"] and out.linear_rgb == m["linear_rgb"], (tag, idx)
        tol = 0.0 if tag in ("over", "in") else 1e-15  # device pow vs numpy pow: last-bit differences
        assert_close64(out.image, g[f"{idx}_out"], atol=tol, what=f"{tag} {idx}")
    with pytest.raises(ValueError):
        S.Layer.compose(ins * 2, 99)
    assert S.Layer.compose([]) is None


# ------------------------------------------------------------------------------------------
# scenes
# ------------------------------------------------------------------------------------------
def _render_dump(S, name, tag):
    from svgrasterize_amd import scenedump

    scene, info, z = scenedump.load_scene(os.path.join(GOLDEN, f"scene_{name}.npz"))
    r = next(r for r in info["renders"] if r["tag"] == tag)
    tr = S.Transform().matrix(0, 1, 0, 1, 0, 0).scale(r["scale"])
    hh, ww = r["size"]
    return scene, z, r, tr, hh, ww


@pytest.mark.parametrize("tag", ["s128", "s256"])
def test_tiger_small(S, tag):
    scene, z, r, tr, hh, ww = _render_dump(S, "tiger", tag)
    res = scene.render(tr, viewport=[0, 0, hh, ww], linear_rgb=False)
    layer, _hull = res
    assert [int(v) for v in layer.offset] == r["layer_offset"]
    assert layer.image.shape == z[f"{tag}_layer"].shape
    assert_close64(layer.image, z[f"{tag}_layer"], atol=1e-10, what="tiger group layer")
    canvas32 = layer.to_canvas_f32(hh, ww)
    assert_f32_1ulp(canvas32, z[f"{tag}_canvas"], what="tiger canvas f32")
    # production entry (one batch straight to a float32 canvas)
    img, st = S.render_canvas(scene, tr, [0, 0, hh, ww], linear_rgb=False)
    assert_f32_1ulp(img, z[f"{tag}_canvas"], what="tiger render_canvas")
    # integer bboxes of every leaf are bit-exact
    from svgrasterize_amd.scene import build_batch

    leaves = scene.leaves(tr, linear_rgb=False)
    batch = build_batch(leaves, [0, 0, hh, ww])
    batch.plan()
    bb = batch.bboxes().astype(np.int64)
    ref_bb = z[f"{tag}_leaf_bbox"]
    empty = ref_bb[:, 2] < 0
    assert np.array_equal(bb[~empty], ref_bb[~empty])
    assert (bb[empty, 2] <= 0).all() or (bb[empty, 3] <= 0).all()


@pytest.mark.parametrize("name,tag", [("tiger", "s128"), ("material", "s256"), ("icons", "s286")])
def test_a_second_render_of_an_unchanged_scene_reuses_the_first_ones_plans(S, name, tag, monkeypatch):
    """Scene.render retains the leaf analysis and the built + planned batches of a (scene, transform, viewport) between
    renders (scene._Retained): the second and third render give the first one's picture, with the cache and without it."""
    from svgrasterize_amd import scene as scene_mod, scenedump

    sc, info, z = scenedump.load_scene(os.path.join(GOLDEN, f"scene_{name}.npz"))
    r = next((r for r in info["renders"] if r["tag"] == tag), info["renders"][0])
    tr = S.Transform().matrix(0, 1, 0, 1, 0, 0).scale(r["scale"])
    hh, ww = r["size"]
    S.clear_render_cache()
    monkeypatch.setattr(scene_mod, "_RETAINED_MAX", 4)   # (whatever $SVGR_RENDER_CACHE says)
    shots = []
    for _ in range(3):
        layer, hull = sc.render(tr, viewport=[0, 0, hh, ww], linear_rgb=False)
        shots.append((np.array(layer.image), tuple(int(v) for v in layer.offset), np.array(hull.points)))
    assert len(scene_mod._RETAINED) == 1
    st = next(iter(scene_mod._RETAINED.values()))
    assert st.scene is sc and (st.run_plans or st.fill_plans or st.jobs)
    S.clear_render_cache()
    assert not scene_mod._RETAINED
    monkeypatch.setattr(scene_mod, "_RETAINED_MAX", 0)
    layer, hull = sc.render(tr, viewport=[0, 0, hh, ww], linear_rgb=False)
    shots.append((np.array(layer.image), tuple(int(v) for v in layer.offset), np.array(hull.points)))
    assert not scene_mod._RETAINED
    for img, off, pts in shots[1:]:
        assert off == shots[0][1] and img.shape == shots[0][0].shape
        assert np.abs(img - shots[0][0]).max() <= 1e-12   # (the order of the LDS atomics: double rounding)
        assert np.array_equal(pts, shots[0][2])


def _two_blobs(S, colour):
    a = S.Path.from_svg("M10,10 C40,0 70,20 60,50 C50,80 20,70 10,40 Z")
    a = S.Path([[(t, np.array(pts, dtype=np.float64)) for t, pts in sub] for sub in a.subpaths])   # (segments as arrays, S:899-907)
    b = S.Path.from_svg("M30,30 L90,35 L85,90 L35,85 Z")
    return S.Scene.group([S.Scene.fill(a, colour, None), S.Scene.fill(b, np.array([0.1, 0.2, 0.6, 0.7]), "evenodd"),
                          S.Scene.fill(a, np.array([0.0, 0.3, 0.0, 0.3]), None).transform(S.Transform().translate(20, 15))]), a, b


def test_the_retained_render_notices_arrays_edited_in_place(S, monkeypatch):
    """The reference keeps nothing between renders (S:649-752): a paint edited in place is drawn with its new values.  The
    (opt-in) retained entry of a scene is guarded by the bytes of the scene's paint arrays: after an in-place edit the next
    render is a cold one, and its picture is the new one.  (Path geometry is a value: `Path.packed` has kept a copy since
    round 1, with or without the cache.)"""
    from svgrasterize_amd import scene as scene_mod

    monkeypatch.setattr(scene_mod, "_RETAINED_MAX", 4)
    monkeypatch.setattr(scene_mod, "_RETAINED_TRUST", False)
    S.clear_render_cache()
    tr = S.Transform().matrix(0, 1, 0, 1, 0, 0)
    vp = [0, 0, 128, 128]
    colour = np.array([0.8, 0.1, 0.1, 0.9])
    scene, _a, _b = _two_blobs(S, colour)
    first = np.array(scene.render(tr, viewport=vp)[0].image)
    again = np.array(scene.render(tr, viewport=vp)[0].image)      # (warm: the retained batches)
    assert np.abs(first - again).max() <= 1e-12
    # a paint edited in place
    colour[:] = [0.1, 0.7, 0.2, 0.8]
    got = scene.render(tr, viewport=vp)[0]
    fresh_scene, _a2, _b2 = _two_blobs(S, colour.copy())
    S.clear_render_cache()
    want = fresh_scene.render(tr, viewport=vp)[0]
    assert tuple(got.offset) == tuple(want.offset) and got.image.shape == want.image.shape
    assert np.abs(np.array(got.image) - np.array(want.image)).max() <= 1e-12
    assert np.abs(np.array(got.image) - first).max() > 0.05        # (and it is not the old picture)
    S.clear_render_cache()


def test_hulls_outlive_the_retained_entry_that_built_their_batch(S, monkeypatch):
    """`Scene.render` returns a LAZY hull: it reads its batch's edges on first use.  Clearing the cache (or evicting the
    entry) only drops references; the batch lives until the hull does (it used to be freed under the hull)."""
    from svgrasterize_amd import scene as scene_mod

    monkeypatch.setattr(scene_mod, "_RETAINED_MAX", 1)
    S.clear_render_cache()
    tr = S.Transform().matrix(0, 1, 0, 1, 0, 0)
    vp = [0, 0, 128, 128]
    scene, _a, _b = _two_blobs(S, np.array([0.8, 0.1, 0.1, 0.9]))
    _layer, hull = scene.render(tr, viewport=vp)
    other, _a2, _b2 = _two_blobs(S, np.array([0.3, 0.3, 0.1, 0.5]))
    other.render(tr.translate(1, 1), viewport=vp)                   # (evicts the first entry: the cache holds one)
    S.clear_render_cache()
    pts = np.array(hull.points)                                     # first use of the hull: after eviction AND clear
    _layer2, hull2 = scene.render(tr, viewport=vp)
    assert pts.shape[0] >= 3 and np.array_equal(pts, np.array(hull2.points))
    S.clear_render_cache()


def test_runs_built_on_demand_do_not_pile_up_in_a_retained_entry(S, monkeypatch):
    """A run the pre-pass does not see (inside an objectBoundingBox mask: its transform comes from the target's hull) is built
    while the walk runs, from leaves that are made afresh in every render: its key never comes back.  The retained entry keeps
    such a run for one render at most (the entry used to grow by one planned batch per render)."""
    from svgrasterize_amd import scene as scene_mod

    monkeypatch.setattr(scene_mod, "_RETAINED_MAX", 4)
    S.clear_render_cache()
    tr = S.Transform().matrix(0, 1, 0, 1, 0, 0)
    vp = [0, 0, 128, 128]
    scene, a, b = _two_blobs(S, np.array([0.8, 0.1, 0.1, 0.9]))
    unit = S.Path.from_svg("M0.1,0.1 L0.9,0.1 L0.9,0.9 L0.1,0.9 Z")
    unit2 = S.Path.from_svg("M0.3,0.2 L0.8,0.3 L0.6,0.8 Z")
    mask_scene = S.Scene.group([S.Scene.fill(unit, np.array([1.0, 1.0, 1.0, 1.0]), None), S.Scene.fill(unit2, np.array([0.2, 0.2, 0.2, 1.0]), None)])
    doc = S.Scene.group([scene, S.Scene.fill(b, np.array([0.5, 0.1, 0.5, 0.8]), None).mask(mask_scene, True), scene.transform(S.Transform().translate(3, 2))])
    sizes, shots = [], []
    for _ in range(5):
        layer, _hull = doc.render(tr, viewport=vp)
        shots.append(np.array(layer.image))
        st = next(iter(scene_mod._RETAINED.values()))
        sizes.append(len(st.run_plans))
    assert max(sizes[1:]) == min(sizes[1:]), sizes                  # stable from the second render on
    for img in shots[1:]:
        assert np.abs(img - shots[0]).max() <= 1e-12
    S.clear_render_cache()


def test_prompt_text_outlines(S):
    """demo/prompt.svg at width 256: glyph outlines set by the reference's fonts, a 9 x 256 strip."""
    scene, z, r, tr, hh, ww = _render_dump(S, "prompt", "s9")
    layer, _ = scene.render(tr, viewport=[0, 0, hh, ww], linear_rgb=False)
    assert [int(v) for v in layer.offset] == r["layer_offset"]
    assert_close64(layer.image, z["s9_layer"], atol=1e-10, what="prompt group layer")
    assert_f32_1ulp(layer.to_canvas_f32(hh, ww), z["s9_canvas"], what="prompt canvas")


def test_material_small_clips(S):
    """material-design: 935 clip paths -> per-node route (CLIP = mask IN image) + batched runs."""
    scene, z, r, tr, hh, ww = _render_dump(S, "material", "s256")
    layer, _ = scene.render(tr, viewport=[0, 0, hh, ww], linear_rgb=False)
    assert [int(v) for v in layer.offset] == r["layer_offset"]
    assert_close64(layer.image, z["s256_layer"], atol=1e-10, what="material group layer")
    assert_f32_1ulp(layer.to_canvas_f32(hh, ww), z["s256_canvas"], what="material canvas")


def test_tiger_crop_viewport(S):
    from svgrasterize_amd import scenedump

    scene, info, z = scenedump.load_scene(os.path.join(GOLDEN, "scene_tiger.npz"))
    vp = info["crop"]["viewport"]
    tr = S.Transform().matrix(0, 1, 0, 1, 0, 0)
    layer, _ = scene.render(tr, viewport=vp, linear_rgb=False)
    assert [int(v) for v in layer.offset] == info["crop"]["offset"]
    assert_close64(layer.image, z["crop_layer"], atol=1e-10, what="tiger crop")


def test_fifty_thousand_small_paths_vs_oracle(S, orc):
    """Path count at the other extreme from the bench scene: 50 000 paths of a dozen pixels each (lines, quads turned
    cubics, both fill rules, many per tile: the tile lists run in batches of 32), one batch, against the oracle."""
    from svgrasterize_amd import _abi

    n, size = 50_000, 1024
    rng = np.random.default_rng(2024)
    c = rng.uniform(-4, size + 4, (n, 1, 2))
    ang = np.sort(rng.uniform(0, 2 * np.pi, (n, 3)), axis=1)
    tri = c + rng.uniform(2, 9, (n, 3, 1)) * np.stack([np.cos(ang), np.sin(ang)], axis=-1)  # (n, 3, 2) corners, (x, y)
    segs = np.zeros((n, 3, 8))
    kinds = np.zeros((n, 3), dtype=np.uint8)
    for k in range(3):
        a, b = tri[:, k], tri[:, (k + 1) % 3]
        curved = (np.arange(n) + k) % 4 == 0
        segs[:, k, 0:2], segs[:, k, 2:4] = a, b  # a line: two points
        bulge = 0.5 * (a + b) + rng.uniform(-2, 2, (n, 2))
        segs[curved, k, 2:4], segs[curved, k, 4:6], segs[curved, k, 6:8] = (a + 2 * (bulge - a) / 3)[curved], (b + 2 * (bulge - b) / 3)[curved], b[curved]
        kinds[curved, k] = 1
    segs, kinds = segs.reshape(-1, 8), kinds.reshape(-1)
    off = np.arange(n + 1, dtype=np.int64) * 3
    alpha = rng.uniform(0.2, 1.0, (n, 1))
    paint = np.concatenate([rng.uniform(0, 1, (n, 3)) * alpha, alpha], axis=1)
    rule = (np.arange(n) % 5 == 0).astype(np.uint8)
    m6 = np.tile(np.array([0.0, 1, 0, 1, 0, 0]), (n, 1))
    vp = [0, 0, size, size]
    pres = segs.copy()
    pres[:, 0::2], pres[:, 1::2] = segs[:, 1::2], segs[:, 0::2]  # the oracle takes presentation space (row, col)
    ref, P, _E = orc.render_solid(pres, kinds, off, rule, paint, vp, clip01=True)
    ctx = S.Context.get()
    batch = _abi.Batch(ctx, segs, kinds, off, m6, rule, paint, viewport=vp)
    st = batch.plan()
    assert st.path_pixels == P
    out = ctx.alloc(size * size * 32)
    batch.render(out, _abi.OUT_CANVAS_F64, _abi.RENDER_CLIP01)
    assert_close64(out.download((size, size, 4), np.float64), ref, atol=1e-10, what="50k small paths")


def test_one_path_of_twenty_thousand_spikes_vs_oracle(S, orc):
    """One path, 40 000 line segments, every band crossed by thousands of its edges: the (path, band) record lists run
    far past one prefetch block (the tail is read straight from HBM), evenodd winding over hundreds of overlaps."""
    from svgrasterize_amd import _abi

    size, spikes = 1536, 20_000
    rng = np.random.default_rng(7)
    ang = np.linspace(0, 2 * np.pi, 2 * spikes, endpoint=False) + rng.uniform(0, 1e-3, 2 * spikes)
    rad = np.where(np.arange(2 * spikes) % 2 == 0, rng.uniform(500, 760, 2 * spikes), rng.uniform(5, 200, 2 * spikes))
    pts = np.stack([size / 2 + rad * np.cos(ang), size / 2 + rad * np.sin(ang)], axis=1)  # (x, y)
    segs = np.zeros((2 * spikes, 8))
    segs[:, 0:2], segs[:, 2:4] = pts, np.roll(pts, -1, axis=0)
    kinds = np.zeros(2 * spikes, dtype=np.uint8)
    off = np.array([0, 2 * spikes], dtype=np.int64)
    for rule in (1, 0):
        paint = np.array([[0.3, 0.1, 0.45, 0.6]])
        vp = [0, 0, size, size]
        pres = segs.copy()
        pres[:, 0::2], pres[:, 1::2] = segs[:, 1::2], segs[:, 0::2]
        ref, P, _E = orc.render_solid(pres, kinds, off, np.array([rule], dtype=np.uint8), paint, vp, clip01=True)
        ctx = S.Context.get()
        batch = _abi.Batch(ctx, segs, kinds, off, np.array([[0.0, 1, 0, 1, 0, 0]]), np.array([rule], dtype=np.uint8), paint, viewport=vp)
        st = batch.plan()
        assert st.path_pixels == P
        out = ctx.alloc(size * size * 32)
        out.zero()
        batch.render(out, _abi.OUT_CANVAS_F64, _abi.RENDER_CLIP01)
        # hundreds of windings cancel along a row: the blocked sums differ from the sequential cumsum by their rounding
        # (double; the measured maximum is recorded next to the float32 contract counts).  The product contract is the
        # float32 one, and it holds with nothing outside it.
        got = out.download((size, size, 4), np.float64)
        c = f32_contract_counts(got.astype(np.float32), ref, f"spikes, rule {rule}: f64 max abs err {float(np.abs(got - ref).max()):.3e}")
        assert c["bad"] == 0, c
        assert_close64(got, ref, atol=5e-10, what=f"spikes, rule {rule}")


# ------------------------------------------------------------------------------------------
# synthetic scene vs the CPU oracle (same seeded input)
# ------------------------------------------------------------------------------------------
@pytest.mark.parametrize("size,n", [(256, 48), (700, 300)])
def test_synthetic_vs_oracle(S, orc, size, n):
    from svgrasterize_amd import _abi, synth

    sc = synth.make_scene(size, n)
    ref, P, E = orc.render_solid(synth.presentation_segs(sc), sc["seg_kind"], sc["path_seg_off"], sc["path_rule"],
                                 sc["path_paint"], sc["viewport"], clip01=True)
    ctx = S.Context.get()
    batch = _abi.Batch(ctx, sc["segs"], sc["seg_kind"], sc["path_seg_off"], sc["path_m6"], sc["path_rule"], sc["path_paint"],
                       viewport=sc["viewport"])
    st = batch.plan()
    assert st.path_pixels == P and 0 < st.n_edges <= E  # the device drops edges that lie wholly above / below the viewport
    out64 = ctx.alloc(size * size * 32)
    batch.render(out64, _abi.OUT_CANVAS_F64, _abi.RENDER_CLIP01)
    assert_close64(out64.download((size, size, 4), np.float64), ref, atol=1e-10, what="synthetic f64")
    out32 = ctx.alloc(size * size * 16)
    batch.render(out32, _abi.OUT_CANVAS_F32, _abi.RENDER_CLIP01)
    assert_f32_1ulp(out32.download((size, size, 4), np.float32), ref, what="synthetic f32")


@pytest.mark.parametrize("key", ["s256_n48", "s700_n300"])
def test_synthetic_vs_the_reference_render(S, key):
    """The bench generator's scene drawn by the reference itself (tests/golden/synth_kat.npz, oracle/gen_golden.py --only synth):
    the float32 canvas of the production kernel within 1 ULP, the double canvas within 1e-10."""
    import json

    from svgrasterize_amd import _abi, synth

    g = load("synth_kat.npz")
    m = next(x for x in json.loads(str(g["meta"])) if x["key"] == key)
    size, n = m["size"], m["paths"]
    ref = g[key + "_canvas"]
    sc = synth.make_scene(size, n)
    ctx = S.Context.get()
    batch = _abi.Batch(ctx, sc["segs"], sc["seg_kind"], sc["path_seg_off"], sc["path_m6"], sc["path_rule"], sc["path_paint"],
                       viewport=sc["viewport"])
    batch.plan()
    out32 = ctx.alloc(size * size * 16)
    batch.render(out32, _abi.OUT_CANVAS_F32, _abi.RENDER_CLIP01)
    assert_f32_1ulp(out32.download((size, size, 4), np.float32), ref, what="synthetic f32 vs the reference")
    out64 = ctx.alloc(size * size * 32)
    batch.render(out64, _abi.OUT_CANVAS_F64, _abi.RENDER_CLIP01)
    assert_close64(out64.download((size, size, 4), np.float64), ref, atol=1e-10, what="synthetic f64 vs the reference")


@pytest.mark.parametrize("world,strip", [(3, 1), (2, 4), (4, 2)])
def test_band_sharding_matches_full(S, world, strip):
    """Rows rendered by 'rank r of N' (interleaved strips of bands, geometry culled to what reaches them)
    equal the same rows of the full render: identical edges and bboxes for every path the rank keeps (a path
    that cannot reach the rank's rows is dropped whole and reports an empty bbox), so only summation order differs."""
    from svgrasterize_amd import _abi, dist as sdist, synth

    size, n = 300, 80
    sc = synth.make_scene(size, n)
    ctx = S.Context.get()
    batch = _abi.Batch(ctx, sc["segs"], sc["seg_kind"], sc["path_seg_off"], sc["path_m6"], sc["path_rule"], sc["path_paint"],
                       viewport=sc["viewport"])
    st_full = batch.plan()
    full = ctx.alloc(size * size * 32)
    batch.render(full, _abi.OUT_CANVAS_F64)
    full = full.download((size, size, 4), np.float64)
    full_bb = batch.bboxes()
    tr = _abi.tile_rows()
    parts, edges_kept = [], 0
    for rank in range(world):
        batch.set_bands(rank, world, strip)
        st = batch.plan()
        bb = batch.bboxes()
        kept = bb[:, 2] > 0
        assert np.array_equal(bb[kept], full_bb[kept]) and st.path_pixels <= st_full.path_pixels  # kept paths: the global bbox
        mine = np.zeros(size + 2, dtype=bool)
        for r0, r1 in sdist.owned_row_ranges(size, tr, rank, world, strip):
            mine[r0:r1] = True
        for q in np.nonzero(~kept & (full_bb[:, 2] > 0))[0]:  # dropped: no edge row (bbox minus its 1-px margins) is ours
            assert not mine[full_bb[q, 0] + 1: full_bb[q, 0] + full_bb[q, 2] - 1].any(), f"rank {rank} dropped path {q}"
        edges_kept += st.n_edges
        rows = batch.owned_rows()
        assert rows == len(sdist.owned_bands(size, tr, rank, world, strip)) * tr
        part = ctx.alloc(rows * size * 32)
        batch.render(part, _abi.OUT_CANVAS_F64)
        part = part.download((rows, size, 4), np.float64)
        parts.append(part)
        for k, (r0, r1) in enumerate(sdist.owned_row_ranges(size, tr, rank, world, strip)):
            assert_close64(part[k * tr: k * tr + (r1 - r0)], full[r0:r1], atol=1e-12, what=f"rank {rank} rows {r0}:{r1}")
    assert_close64(sdist.assemble(parts, size, tr, strip), full, atol=1e-12, what="assembled")
    assert edges_kept < world * st_full.n_edges  # border edges are duplicated, everything else is shared out
    batch.set_bands(0, 1, 1)


def test_row_block_of_a_tall_scene_matches_the_full_render(S):
    """Weak-scaling decomposition: a GPU that renders rows [r0, r1) of a tall drawing from only the paths that reach
    them (synth.rows_subscene) gets the same pixels as those rows of the full render."""
    from svgrasterize_amd import _abi, synth

    size, n, blocks = 192, 40, 3
    tall = synth.make_tall_scene(size, n, blocks)
    assert tall["viewport"] == (0, 0, size * blocks, size) and len(tall["path_seg_off"]) == n * blocks + 1
    ctx = S.Context.get()

    def render(sc):
        b = _abi.Batch(ctx, sc["segs"], sc["seg_kind"], sc["path_seg_off"], sc["path_m6"], sc["path_rule"], sc["path_paint"],
                       viewport=sc["viewport"])
        b.plan()
        rows, cols = sc["viewport"][2], sc["viewport"][3]
        out = ctx.alloc(rows * cols * 32)
        b.render(out, _abi.OUT_CANVAS_F64)
        return out.download((rows, cols, 4), np.float64)

    full = render(tall)
    kept_total = 0
    for k in range(blocks):
        sub, kept = synth.rows_subscene(tall, k * size, (k + 1) * size)
        kept_total += len(kept)
        assert 0 < len(kept) < n * blocks and np.all(np.diff(kept) > 0)  # a proper subset, paint order kept
        assert_close64(render(sub), full[k * size:(k + 1) * size], atol=1e-12, what=f"row block {k}")
    assert kept_total > n * blocks  # border paths are rendered by both neighbours
