"""Gradient paint servers (S:1021-1047, 1544-1695) and Gaussian blur (S:106-118, 1890-1944): the numpy
oracle against the reference fixtures (CPU), and the HIP kernels against the same fixtures (GPU)."""
import json
import os

import numpy as np
import pytest

from oracle import oracle as orc
from tests.util import GOLDEN, assert_close64, assert_f32_1ulp, f32_contract_counts, load


def _meta():
    g = load("gradient_blur_kat.npz")
    return g, json.loads(str(g["meta"]))


def _solid(paint, linear_rgb):
    return orc.paint_for_fill(paint, linear_rgb)


# ------------------------------------------------------------------------------------------ CPU
def test_oracle_gradients_match_reference():
    g, m = _meta()
    for idx, c in enumerate(m["grad"]):
        tr = g[f"{idx}_tr"]
        res = orc.path_mask(g[f"{idx}_lines"], g[f"{idx}_cubics"], tr, None, c["viewport"])
        mask, off, edges = res
        assert list(off) == c["offset"]
        user_m = np.linalg.inv(tr)
        if c["bbox_units"]:
            pts = (np.linalg.inv(tr)[:2, :2] @ _hull(edges).T).T + np.linalg.inv(tr)[:2, 2]
            x, y = pts.min(axis=0)
            w, h = pts.max(axis=0) - pts.min(axis=0)
            user_m = np.linalg.inv(tr @ np.array([[1, 0, x], [0, 1, y], [0, 0, 1.0]]) @ np.array([[w, 0, 0], [0, h, 0], [0, 0, 1.0]]))
        lin = c["layer_linear_rgb"]
        stops = np.array([_solid(col, lin) for col in g[f"{idx}_stop_col"]])
        gt = np.linalg.inv(g[f"{idx}_gt"]) if c["has_gt"] else None
        kw = dict(p0=g[f"{idx}_p0"], p1=g[f"{idx}_p1"]) if c["kind"] == "linear" else dict(
            center=g[f"{idx}_center"], radius=float(g[f"{idx}_radius"]),
            fcenter=g[f"{idx}_fcenter"] if c.get("has_focal") else None,
            fradius=float(g[f"{idx}_fradius"]) if c.get("has_focal") else None)
        img = orc.gradient_image(c["kind"], (off[0], off[1]) + mask.shape, user_m, gt, c["spread"], g[f"{idx}_stop_off"], stops, **kw)
        assert_close64(img * mask[..., None], g[f"{idx}_image"], atol=1e-12, what=f"gradient {idx}")


def _hull(edges):
    from svgrasterize_amd.geometry import ConvexHull

    return np.array(ConvexHull(np.asarray(edges)).points)


def test_oracle_blur_matches_reference():
    g, m = _meta()
    for j, b in enumerate(m["blur"]):
        if b["noop"]:
            assert np.array_equal(g[f"b{j}_out"], g[f"b{j}_in"])
            continue
        img = orc.convert(g[f"b{j}_in"], True, b["in_linear_rgb"], False, True)
        out = orc.convolve_full(img, g[f"b{j}_kernel"])
        assert_close64(out, g[f"b{j}_out"], atol=1e-14, what=f"blur {j}")


# ------------------------------------------------------------------------------------------ GPU
@pytest.mark.gpu
def test_gpu_gradient_fills():
    import svgrasterize_amd as S

    g, m = _meta()
    for idx, c in enumerate(m["grad"]):
        path = S.Path.from_arrays(g[f"{idx}_lines"], g[f"{idx}_cubics"])
        stops = [(float(o), col) for o, col in zip(g[f"{idx}_stop_off"], g[f"{idx}_stop_col"])]
        gt = S.Transform(g[f"{idx}_gt"]) if c["has_gt"] else None
        if c["kind"] == "linear":
            paint = S.GradLinear(g[f"{idx}_p0"], g[f"{idx}_p1"], stops, gt, c["spread"], c["bbox_units"], c["paint_linear_rgb"])
        else:
            paint = S.GradRadial(g[f"{idx}_center"], float(g[f"{idx}_radius"]),
                                 g[f"{idx}_fcenter"] if c.get("has_focal") else None,
                                 float(g[f"{idx}_fradius"]) if c.get("has_focal") else None,
                                 stops, gt, c["spread"], c["bbox_units"], c["paint_linear_rgb"])
        layer, _hull_ = path.fill(S.Transform(g[f"{idx}_tr"]), paint, viewport=c["viewport"], linear_rgb=c["linear_rgb"])
        assert [int(v) for v in layer.offset] == c["offset"]
        assert layer.linear_rgb == c["layer_linear_rgb"] and layer.pre_alpha
        assert_close64(layer.image, g[f"{idx}_image"], atol=1e-11, what=f"gradient {idx} ({c['kind']}, {c['spread']})")
        assert_f32_1ulp(layer.image.astype(np.float32), g[f"{idx}_image"], what=f"gradient {idx}")
        # Grad*.fill on the caller's own coordinates (S:1553 / S:1577), both colour spaces
        pts = g[f"{idx}_eval_pts"]
        assert_close64(paint.fill(pts, linear_rgb=True), g[f"{idx}_eval_lin"], atol=1e-13, what=f"gradient {idx} eval linear")
        assert_close64(paint.fill(pts, linear_rgb=False), g[f"{idx}_eval_srgb"], atol=1e-13, what=f"gradient {idx} eval sRGB")
    assert paint.fill(np.zeros((0, 2))).shape == (0, 4)


@pytest.mark.gpu
def test_gpu_blur():
    import svgrasterize_amd as S

    g, m = _meta()
    for j, b in enumerate(m["blur"]):
        layer = S.Layer(g[f"b{j}_in"].copy(), tuple(b["in_offset"]), True, b["in_linear_rgb"])
        flt = S.Filter.empty().blur(b["sigma"][0], b["sigma"][1])
        out = flt(S.Transform(g[f"b{j}_tr"]), layer)
        assert [int(v) for v in out.offset] == b["out_offset"], j
        assert (out.pre_alpha, out.linear_rgb) == (b["out_pre_alpha"], b["out_linear_rgb"]), j
        assert_close64(out.image, g[f"b{j}_out"], atol=1e-13, what=f"blur {j}")


@pytest.mark.gpu
def test_gpu_icons_scene():
    """demo/icons.svg at native size: 431 gradient fills, 65 clips, 123 opacity groups, 37 blurs."""
    import svgrasterize_amd as S
    from svgrasterize_amd import scenedump

    scene, info, z = scenedump.load_scene(os.path.join(GOLDEN, "scene_icons.npz"))
    r = info["renders"][0]
    hh, ww = r["size"]
    tr = S.Transform().matrix(0, 1, 0, 1, 0, 0).scale(r["scale"])
    layer, _ = scene.render(tr, viewport=[0, 0, hh, ww], linear_rgb=False)
    assert [int(v) for v in layer.offset] == r["layer_offset"]
    canvas = layer.to_canvas_f32(hh, ww)
    # blur goes through FFT in the reference (scipy picks it): absolute noise ~1e-16 there, so the float32 contract holds
    # except where that noise decides a float32 rounding tie or the 1e-6 coverage cut; the counts are recorded per class
    c = f32_contract_counts(canvas, z["s286_canvas"], "icons.svg @1114x286")
    assert c["bad"] == 0 and c["max_err"] < 1e-6, c  # (no value outside the 1-ULP contract: no tie / cut allowance)


@pytest.mark.gpu
def test_gpu_blur_separable_equals_direct(monkeypatch):
    """feGaussianBlur kernels of axis-aligned transforms are rank 1: svgr_layer_convolve runs them as two 1-D passes.
    Same result as the direct 2-D stencil to double rounding; a rotated blur (not separable) takes the 2-D stencil."""
    import svgrasterize_amd as S
    from svgrasterize_amd.filters import blur_kernel

    rng = np.random.default_rng(5)
    img = rng.uniform(0, 1, (57, 83, 4))
    layer = S.Layer(img, (4, 9), pre_alpha=False, linear_rgb=True)
    swap = S.Transform().matrix(0, 1, 0, 1, 0, 0)
    for tr, expect_sep in [(swap.scale(3.0), True), (swap.scale(2.0, 5.0), True), (swap.rotate(0.4).scale(3.0), False)]:
        kern = blur_kernel(tr, (1.7, 2.9))
        assert kern is not None and kern.shape[0] > 3
        u, v = kern.sum(1), kern.sum(0)
        rank1 = np.abs(kern - np.outer(u, v) / kern.sum()).max() <= 8 * np.finfo(float).eps * kern.max()
        assert rank1 == expect_sep
        monkeypatch.delenv("SVGR_BLUR_DIRECT", raising=False)
        auto = layer.convolve(kern)
        monkeypatch.setenv("SVGR_BLUR_DIRECT", "1")
        direct = layer.convolve(kern)
        monkeypatch.delenv("SVGR_BLUR_DIRECT", raising=False)
        assert auto.offset == direct.offset and auto.image.shape == direct.image.shape
        np.testing.assert_allclose(auto.image, direct.image, rtol=0, atol=4e-16 * kern.size ** 0.5 + 1e-15)
        from scipy.signal import convolve as sconv
        want = sconv(img, kern[..., None], mode="full", method="direct")
        np.testing.assert_allclose(auto.image, want, rtol=0, atol=1e-14)


_ICONSET = sorted(os.path.basename(p)[len("scene_"):-len(".npz")] for p in __import__("glob").glob(os.path.join(GOLDEN, "scene_icon_*.npz")))


@pytest.mark.gpu
@pytest.mark.parametrize("name", _ICONSET)
def test_gpu_demo_icon_thumbnails(name):
    """The reference's other demo icons (firefox, inkscape, kde, office, python, ...) at 192 px: whatever mix of
    gradients, clips, filters, opacity groups and strokes real documents bring, against the reference's canvas."""
    import svgrasterize_amd as S
    from svgrasterize_amd import scenedump

    scene, info, z = scenedump.load_scene(os.path.join(GOLDEN, f"scene_{name}.npz"))
    r = info["renders"][0]
    hh, ww = r["size"]
    tr = S.Transform().matrix(0, 1, 0, 1, 0, 0).scale(r["scale"])
    layer, _ = scene.render(tr, viewport=[0, 0, hh, ww], linear_rgb=False)
    assert [int(v) for v in layer.offset] == r["layer_offset"]
    canvas = layer.to_canvas_f32(hh, ww)
    c = f32_contract_counts(canvas, z[f"{r['tag']}_canvas"], f"{name} @192")  # (classes: see tests/util.py; counts on file)
    assert c["bad"] == 0 and c["max_err"] < 1e-6, c  # (no value outside the 1-ULP contract: no tie / cut allowance)


@pytest.mark.gpu
def test_gpu_gradients_with_many_stops():
    """33 to 200 stops (the reference loops over any number, S:1671-1683): evaluation at points in both colour spaces and a
    filled path, against the reference (tests/golden/gradlong_kat.npz, gen_golden.py --only gradlong)."""
    import json

    import svgrasterize_amd as S

    z = np.load(os.path.join(GOLDEN, "gradlong_kat.npz"))
    for i, m in enumerate(json.loads(str(z["meta"]))):
        stops = [(float(o), c) for o, c in zip(z[f"{i}_off"], z[f"{i}_rgba"])]
        assert len(stops) == m["n"] > 32
        if m["kind"] == "linear":
            paint = S.GradLinear(np.array([2.0, 3.0]), np.array([40.0, 25.0]), stops, None, m["spread"], False, None)
        else:
            paint = S.GradRadial(np.array([20.0, 18.0]), 17.0, None, None, stops, None, m["spread"], False, None)
        assert_close64(paint.fill(z[f"{i}_pts"], linear_rgb=True), z[f"{i}_lin"], atol=1e-13, what=f"{m} eval linear")
        assert_close64(paint.fill(z[f"{i}_pts"], linear_rgb=False), z[f"{i}_srgb"], atol=1e-13, what=f"{m} eval sRGB")
        layer, _ = S.Path.from_svg("M3,2 H45 V38 H3 Z").fill(S.Transform().matrix(0, 1, 0, 1, 0, 0), paint, linear_rgb=False)
        assert [int(v) for v in layer.offset] == m["offset"]
        assert_close64(layer.image, z[f"{i}_image"], atol=1e-11, what=f"{m} fill")
        assert_f32_1ulp(layer.image.astype(np.float32), z[f"{i}_image"], what=f"{m} fill")


@pytest.mark.gpu
def test_gpu_icons_4096():
    """BASELINE config 5 at its stated size: demo/icons.svg at width 4096 (4096 x 1051; 431 gradient fills, 36 blurs with
    kernels up to 73 x 73, S:1903-1944) against sparse pins of the reference's own render (sha256 of its float32 canvas
    63dd9502a69c5a7c; oracle/gen_golden.py --only icons4096 --full)."""
    import svgrasterize_amd as S
    from svgrasterize_amd import scenedump

    scene, info, z = scenedump.load_scene(os.path.join(GOLDEN, "scene_icons4096.npz"))
    assert info["full"]["sha256_f32"].startswith("63dd9502a69c5a7c")
    h, w = info["full"]["size"]
    tr = S.Transform().matrix(0, 1, 0, 1, 0, 0)
    layer, _ = scene.render(tr, viewport=[0, 0, h, w], linear_rgb=False)
    assert [int(v) for v in layer.offset] == info["full"]["layer_offset"]
    assert list(layer.image.shape) == info["full"]["layer_shape"]
    canvas = layer.to_canvas_f32(h, w)
    c = f32_contract_counts(canvas.reshape(-1, 4)[z["full_idx"]], z["full_val"], f"icons.svg @{w}x{h} ({len(z['full_idx'])} pins)")
    assert c["bad"] == 0 and c["max_err"] < 1e-6, c  # (no value outside the 1-ULP contract: no tie / cut allowance)


@pytest.mark.gpu
def test_gpu_runs_of_a_document_are_planned_together(monkeypatch):
    """Scene.render's pre-pass finds every run of fills the walk is going to flush, builds their batches and plans them
    behind one wait (svgr_batch_plan_many).  The walk must then meet exactly those runs (every pre-planned batch is used,
    none is planned a second time), and the picture must be the one the run-by-run plans give."""
    import svgrasterize_amd as S
    from svgrasterize_amd import _abi, scenedump
    from svgrasterize_amd import scene as sm

    scene, info, _z = scenedump.load_scene(os.path.join(GOLDEN, "scene_icons.npz"))
    r = info["renders"][0]
    hh, ww = r["size"]
    tr = S.Transform().matrix(0, 1, 0, 1, 0, 0).scale(r["scale"])

    seen = {"planned": 0, "hits": 0, "runs": 0, "single_plans": 0}
    orig_plan_runs, orig_render_run, orig_plan = sm._plan_runs, sm._render_run, _abi.Batch.plan

    def plan_runs(runs, fills, viewport, linear_rgb):
        plans, fill_plans = orig_plan_runs(runs, fills, viewport, linear_rgb)
        seen["planned"], seen["fills"] = len(plans), len(fill_plans)
        return plans, fill_plans

    def render_run(leaves, viewport, linear_rgb):
        seen["runs"] += 1
        if sm.STATE.run_plans and leaves and sm._run_key(leaves, viewport) in sm.STATE.run_plans:
            seen["hits"] += 1
        before = seen["single_plans"]
        out = orig_render_run(leaves, viewport, linear_rgb)
        seen["replanned"] = seen.get("replanned", 0) + (seen["single_plans"] - before)
        return out

    def plan(self):
        seen["single_plans"] += 1
        return orig_plan(self)

    monkeypatch.setattr(sm, "_plan_runs", plan_runs)
    monkeypatch.setattr(sm, "_render_run", render_run)
    monkeypatch.setattr(_abi.Batch, "plan", plan)
    monkeypatch.setattr(sm, "_NODE_RUNS", True)   # (whatever the environment's switches say: the counts below are this route's)
    layer, _ = scene.render(tr, viewport=[0, 0, hh, ww], linear_rgb=False)
    together = layer.to_canvas_f32(hh, ww)
    assert seen["planned"] >= 10, seen                      # (icons.svg: the 37 filter nodes cut the paint order into runs)
    assert seen["hits"] == seen["planned"], seen            # every batch of the pre-pass was met by the walk
    assert seen["replanned"] == 0 or seen["runs"] > seen["hits"], seen
    assert seen["replanned"] == seen["runs"] - seen["hits"], seen   # only runs the pre-pass did not have plan for themselves
    # the fills under filter nodes are runs of their own (round 5, `_NODE_RUNS`; before, the solid ones were planned in the same wait
    # as single-path batches): planned with the others, all picked up, no node-by-node solid fill left
    from svgrasterize_amd import geometry as gm

    assert seen["fills"] == 0 and seen["planned"] >= 40 and gm.STATE.fill_plans is None, seen
    assert seen["single_plans"] <= 3, seen                          # (objectBoundingBox clips and the like plan on demand)

    monkeypatch.setattr(sm, "_plan_runs", lambda runs, fills, viewport, linear_rgb: ({}, {}))
    layer, _ = scene.render(tr, viewport=[0, 0, hh, ww], linear_rgb=False)
    one_by_one = layer.to_canvas_f32(hh, ww)
    assert np.abs(together.astype(np.float64) - one_by_one).max() <= 2.0 ** -23   # (same kernels; LDS atomic order only)


@pytest.mark.gpu
@pytest.mark.parametrize("name,tag", [("icons", "s286"), ("material", "s256"), ("tiger", "s128")])
def test_gpu_runs_sharing_one_batch_draw_what_their_own_batches_draw(name, tag, monkeypatch):
    """The runs of fills of a document share ONE device batch (each in a band range of its own, one geometry pass per
    render, a render window per run: VERDICT r3 #7).  Same layer as with a batch per run: same offset and shape, values to
    1e-10 (a run's rows are moved down by a whole number of bands -- thousands of rows --, so the sub-pixel positions lose
    a dozen of their 50 bits: 2e-12 of a pixel, times a gradient's slope), also on the second -- retained -- render; and
    the document's runs did share.  The reference pins of the scenes (1 ULP of float32) are checked with sharing on."""
    import svgrasterize_amd as S
    from svgrasterize_amd import scenedump
    from svgrasterize_amd import scene as sm

    scene, info, _z = scenedump.load_scene(os.path.join(GOLDEN, f"scene_{name}.npz"))
    r = next((x for x in info["renders"] if x.get("tag") == tag), info["renders"][0])
    hh, ww = r["size"]
    tr = S.Transform().matrix(0, 1, 0, 1, 0, 0).scale(r["scale"])

    def draw():
        S.clear_render_cache()
        out = []
        for _ in range(2):
            layer, hull = scene.render(tr, viewport=[0, 0, hh, ww], linear_rgb=False)
            out.append(([int(v) for v in layer.offset], np.array(layer.image), np.array(hull.points)))
        S.clear_render_cache()
        return out

    monkeypatch.setattr(sm, "_MERGE_RUNS", False)
    alone = draw()
    monkeypatch.setattr(sm, "_MERGE_RUNS", True)
    before = dict(sm.MERGE_STATS)
    shared = draw()
    for (o1, a, h1), (o2, b, h2) in zip(alone, shared):
        assert o1 == o2 and a.shape == b.shape
        assert float(np.abs(a - b).max(initial=0.0)) <= 1e-10   # (measured on icons.svg: 5.6e-12)
        assert h1.shape == h2.shape and float(np.abs(h1 - h2).max(initial=0.0)) <= 1e-9   # (hull points: moved down and back)
    if name == "icons":
        assert sm.MERGE_STATS["runs_sharing"] - before["runs_sharing"] >= 10, sm.MERGE_STATS


@pytest.mark.gpu
def test_gpu_node_by_node_fills_sharing_one_batch_draw_what_their_own_batches_draw(monkeypatch):
    """The solid fills icons.svg draws node by node (children of its filter nodes) share one batch and one launch per render
    (SVGR_OUT_FILLS_F64): the picture is the one their single-path batches give (positions unchanged: to the order of the
    LDS atomics), on the first and on the retained render; and there were such fills."""
    import svgrasterize_amd as S
    from svgrasterize_amd import geometry as gm, scenedump

    scene, info, _z = scenedump.load_scene(os.path.join(GOLDEN, "scene_icons.npz"))
    r = info["renders"][0]
    hh, ww = r["size"]
    tr = S.Transform().matrix(0, 1, 0, 1, 0, 0).scale(r["scale"])
    seen = {"views": 0}
    orig = gm.FillView.layer_buffer

    def layer_buffer(self):
        seen["views"] += 1
        return orig(self)

    monkeypatch.setattr(gm.FillView, "layer_buffer", layer_buffer)

    def draw():
        S.clear_render_cache()
        out = []
        for _ in range(2):
            layer, _hull = scene.render(tr, viewport=[0, 0, hh, ww], linear_rgb=False)
            out.append(([int(v) for v in layer.offset], np.array(layer.image)))
        S.clear_render_cache()
        return out

    from svgrasterize_amd import scene as sm

    monkeypatch.setattr(sm, "_NODE_RUNS", False)   # (round 5 draws these fills as runs of the document's shared batch: the route under test is the one behind the switch)
    monkeypatch.setattr(gm, "_SHARE_FILLS", False)
    alone = draw()
    assert seen["views"] == 0
    monkeypatch.setattr(gm, "_SHARE_FILLS", True)
    shared = draw()
    assert seen["views"] >= 10, seen
    for (o1, a), (o2, b) in zip(alone, shared):
        assert o1 == o2 and a.shape == b.shape
        assert float(np.abs(a - b).max(initial=0.0)) <= 1e-12


@pytest.mark.gpu
def test_gpu_run_windows_drawn_side_by_side_are_the_windows_drawn_one_by_one(monkeypatch):
    """svgr_batch_render_windows (the runs' windows of one render on streams of their own, between two events of the context's
    stream; `SVGR_WINDOW_PREFETCH`): the same layers as one svgr_batch_render_window after the other, twice in a row."""
    import svgrasterize_amd as S
    from svgrasterize_amd import _abi, scenedump
    from svgrasterize_amd import scene as sm

    scene, info, _z = scenedump.load_scene(os.path.join(GOLDEN, "scene_icons.npz"))
    r = info["renders"][0]
    hh, ww = r["size"]
    tr = S.Transform().matrix(0, 1, 0, 1, 0, 0).scale(r["scale"])
    seen = {"calls": 0, "windows": 0}
    orig = _abi.Batch.render_windows

    def render_windows(self, outs, kind, windows, flags=0):
        seen["calls"] += 1
        seen["windows"] += len(outs)
        return orig(self, outs, kind, windows, flags)

    monkeypatch.setattr(_abi.Batch, "render_windows", render_windows)

    def draw():
        S.clear_render_cache()
        out = []
        for _ in range(2):
            layer, _hull = scene.render(tr, viewport=[0, 0, hh, ww], linear_rgb=False)
            out.append(([int(v) for v in layer.offset], np.array(layer.image)))
        S.clear_render_cache()
        return out

    monkeypatch.setattr(sm, "_MERGE_RUNS", True)   # (whatever the environment's switches say)
    monkeypatch.setattr(sm, "_PREFETCH_WINDOWS", False)
    one_by_one = draw()
    assert seen["calls"] == 0
    monkeypatch.setattr(sm, "_PREFETCH_WINDOWS", True)
    together = draw()
    assert seen["calls"] >= 2 and seen["windows"] >= 20, seen
    for (o1, a), (o2, b) in zip(one_by_one, together):
        assert o1 == o2 and a.shape == b.shape
        assert float(np.abs(a - b).max(initial=0.0)) <= 1e-12


@pytest.mark.gpu
def test_gpu_bounding_box_gradients_inside_the_batch_draw_what_they_draw_node_by_node(monkeypatch):
    """objectBoundingBox gradients as batch entries (VERDICT r4 #5b): the frame comes from the extent of the path's flattened
    points between the plan and the first render (svgr_batch_get_extents -> `_resolve_frames`), not from a hull fetched per
    fill.  Same picture as the per-node route (Path.mask, hull.bbox_transform, svgr_gradient_fill, S:1021-1047) -- linear,
    radial and two-circle gradients, pad / repeat / reflect, under a scale + x/y swap, with an opacity on top and inside a
    clipped group.  icons.svg (its dump holds user-space gradients only) is drawn both ways as well, and with its 36 fills under
    filter nodes as runs of the shared batch against the node-by-node route (`_NODE_RUNS`)."""
    import svgrasterize_amd as S
    from svgrasterize_amd import scenedump
    from svgrasterize_amd import scene as sm

    red, blue, green = np.array([1.0, 0.0, 0.0, 1.0]), np.array([0.0, 0.0, 1.0, 1.0]), np.array([0.0, 0.5, 0.0, 0.5])
    stops = [(0.0, red), (0.4, green), (1.0, blue)]

    def blob(cx, cy, r):
        k = 0.5522847498 * r
        return S.Path([[(S.PATH_CUBIC, [[cx + r, cy], [cx + r, cy + k], [cx + k, cy + r], [cx, cy + r]]),
                        (S.PATH_CUBIC, [[cx, cy + r], [cx - k, cy + r], [cx - r, cy + k], [cx - r, cy]]),
                        (S.PATH_CUBIC, [[cx - r, cy], [cx - r, cy - k], [cx - k, cy - r], [cx, cy - r]]),
                        (S.PATH_CUBIC, [[cx, cy - r], [cx + k, cy - r], [cx + r, cy - k], [cx + r, cy]])]])

    lin = S.GradLinear(np.array([0.0, 0.0]), np.array([1.0, 1.0]), stops, None, "pad", True, None)
    lin_t = S.GradLinear(np.array([0.1, 0.0]), np.array([0.6, 0.2]), stops, S.Transform().rotate(0.4), "reflect", True, None)
    rad = S.GradRadial(np.array([0.5, 0.5]), 0.5, None, None, stops, None, "repeat", True, None)
    foc = S.GradRadial(np.array([0.5, 0.5]), 0.45, np.array([0.35, 0.4]), 0.05, stops, None, "pad", True, None)
    user = S.GradLinear(np.array([10.0, 10.0]), np.array([90.0, 40.0]), stops, None, "pad", False, None)
    fill = lambda p, paint: S.Scene.fill(p, paint)  # noqa: E731
    grp = S.Scene.group([fill(blob(150, 60, 30), red), fill(blob(160, 70, 28), rad)])
    scene = S.Scene.group([
        fill(blob(40, 40, 30), lin), fill(blob(90, 50, 35), lin_t).opacity(0.6), fill(blob(60, 100, 38), rad), fill(blob(120, 110, 33), foc),
        fill(blob(30, 120, 25), user), grp.clip(fill(blob(165, 65, 20), blue)), fill(blob(200, 100, 45), lin_t),   # (the last one hangs out of the viewport)
    ])
    tr = S.Transform().matrix(0, 1, 0, 1, 0, 0).scale(2.5, 1.75)
    vp = [0, 0, 300, 520]
    resolved = {"n": 0}
    orig = sm._resolve_frames

    def resolve(batch):
        if getattr(batch, "_frames", None):
            resolved["n"] += len(batch._frames[2])
        return orig(batch)

    monkeypatch.setattr(sm, "_resolve_frames", resolve)

    def draw(sc, t, view):
        S.clear_render_cache()
        layer, _hull = sc.render(t, viewport=view, linear_rgb=False)
        return [int(v) for v in layer.offset], np.array(layer.image)

    monkeypatch.setattr(sm, "_BATCH_BBOX_GRADS", False)
    o1, a = draw(scene, tr, vp)
    assert resolved["n"] == 0
    monkeypatch.setattr(sm, "_BATCH_BBOX_GRADS", True)
    o2, b = draw(scene, tr, vp)
    assert resolved["n"] == 6, resolved
    assert o1 == o2 and a.shape == b.shape and a.any()
    assert float(np.abs(a - b).max()) <= 1e-11
    # icons.svg, first and retained render
    iscene, info, _z = scenedump.load_scene(os.path.join(GOLDEN, "scene_icons.npz"))
    r = info["renders"][0]
    hh, ww = r["size"]
    itr = S.Transform().matrix(0, 1, 0, 1, 0, 0).scale(r["scale"])
    monkeypatch.setattr(sm, "_BATCH_BBOX_GRADS", False)
    o1, a = draw(iscene, itr, [0, 0, hh, ww])
    n0 = resolved["n"]
    monkeypatch.setattr(sm, "_BATCH_BBOX_GRADS", True)
    o2, b = draw(iscene, itr, [0, 0, hh, ww])
    assert resolved["n"] == n0   # (the dump's gradients are all in user space: nothing to resolve, nothing changed)
    assert o1 == o2 and a.shape == b.shape
    assert float(np.abs(a - b).max()) <= 1e-12
    # ... and its fills under filter nodes (gradient and solid, some under an opacity) as runs of the shared batch against node by node
    monkeypatch.setattr(sm, "_NODE_RUNS", False)
    o3, c = draw(iscene, itr, [0, 0, hh, ww])
    assert o3 == o2 and c.shape == b.shape
    assert float(np.abs(c - b).max()) <= 1e-9   # (rows moved by whole bands inside the shared canvas)
    monkeypatch.setattr(sm, "_NODE_RUNS", True)
    S.set_render_cache(2)
    try:
        first = draw(iscene, itr, [0, 0, hh, ww])[1]
        layer, _hull = iscene.render(itr, viewport=[0, 0, hh, ww], linear_rgb=False)   # (retained: the frames are the batch's)
        assert float(np.abs(np.array(layer.image) - first).max()) <= 1e-12
    finally:
        S.set_render_cache(0)


@pytest.mark.gpu
def test_gpu_bounding_box_gradients_in_runs_that_share_a_batch(monkeypatch):
    """objectBoundingBox gradients in runs that SHARE a batch: a blur between two stretches of fills cuts the document
    into runs, `_merge_runs` gives each a range of rows of one tall canvas (its leaves moved down by a whole number of bands), and
    `_resolve_frames` takes a fill's frame from the extent of the MOVED geometry, minus the shift.  (t + shift) - shift is not t in
    double, and the reference takes the hull's box from unshifted points (S:1023-1027, S:2010-2020): the picture against the
    per-node route (`_BATCH_BBOX_GRADS` off: Path.mask, hull.bbox_transform, svgr_gradient_fill) under the float32 1-ULP contract,
    and to 1e-10 in double."""
    import svgrasterize_amd as S
    from svgrasterize_amd import scene as sm

    red, blue, green = np.array([1.0, 0.0, 0.0, 1.0]), np.array([0.0, 0.0, 1.0, 1.0]), np.array([0.0, 0.5, 0.0, 0.5])
    stops = [(0.0, red), (0.4, green), (1.0, blue)]

    def blob(cx, cy, r):
        k = 0.5522847498 * r
        return S.Path([[(S.PATH_CUBIC, [[cx + r, cy], [cx + r, cy + k], [cx + k, cy + r], [cx, cy + r]]),
                        (S.PATH_CUBIC, [[cx, cy + r], [cx - k, cy + r], [cx - r, cy + k], [cx - r, cy]]),
                        (S.PATH_CUBIC, [[cx - r, cy], [cx - r, cy - k], [cx - k, cy - r], [cx, cy - r]]),
                        (S.PATH_CUBIC, [[cx, cy - r], [cx + k, cy - r], [cx + r, cy - k], [cx + r, cy]])]])

    lin = S.GradLinear(np.array([0.0, 0.0]), np.array([1.0, 1.0]), stops, None, "pad", True, None)
    rad = S.GradRadial(np.array([0.5, 0.5]), 0.5, None, None, stops, None, "repeat", True, None)
    foc = S.GradRadial(np.array([0.5, 0.5]), 0.45, np.array([0.35, 0.4]), 0.05, stops, None, "reflect", True, None)
    fill = lambda p, paint: S.Scene.fill(p, paint)  # noqa: E731
    blurred = fill(blob(120.3, 90.7, 22), red).filter(S.Filter.empty().blur(1.5, 1.5))
    runs = []
    for j in range(3):   # three stretches of fills with a blurred node between them: three runs, three ranges of the shared canvas
        dx, dy = 31.37 * j, 17.91 * j
        runs.append([fill(blob(40.21 + dx, 40.6 + dy, 30.13), lin), fill(blob(95.5 + dx, 52.3 + dy, 33.7), rad).opacity(0.7),
                     fill(blob(66.9 + dx, 101.2 + dy, 36.4), foc), fill(blob(150.1 + dx, 120.8 + dy, 20.2), blue)])
    scene = S.Scene.group(runs[0] + [blurred] + runs[1] + [blurred.transform(S.Transform().translate(40, 20))] + runs[2])
    tr = S.Transform().matrix(0, 1, 0, 1, 0, 0).scale(2.25, 1.6)
    vp = [0, 0, 420, 640]
    before = dict(sm.MERGE_STATS)

    def draw():
        S.clear_render_cache()
        layer, _hull = scene.render(tr, viewport=vp, linear_rgb=False)
        return [int(v) for v in layer.offset], np.array(layer.image)

    monkeypatch.setattr(sm, "_BATCH_BBOX_GRADS", False)
    o1, a = draw()
    monkeypatch.setattr(sm, "_BATCH_BBOX_GRADS", True)
    o2, b = draw()
    assert sm.MERGE_STATS["runs_sharing"] - before["runs_sharing"] >= 3, "the runs did not share a batch: nothing was moved"
    assert o1 == o2 and a.shape == b.shape and a.any()
    assert float(np.abs(a - b).max()) <= 1e-10
    assert_f32_1ulp(b.astype(np.float32), a, what="bbox-units gradients in shifted runs vs the per-node route")

