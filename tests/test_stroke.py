"""Path.stroke: the native stroker (csrc/svgr_stroke.cpp, host C++, no GPU) against stroke outlines
produced by the reference itself (tests/golden/stroke_kat.npz, oracle/gen_golden.py --only stroke): every cap x join on
hand-written shapes (polylines, cusps, loops, quads, arcs, degenerate pieces) and every STROKE node of the tiger."""
import json
import os

import numpy as np
import pytest

GOLD = os.path.join(os.path.dirname(__file__), "golden")


@pytest.fixture(scope="module")
def kat():
    z = np.load(os.path.join(GOLD, "stroke_kat.npz"))
    return z, json.loads(str(z["meta"]))


def test_stroke_matches_reference(kat):
    """Same segment kinds, same subpath structure, coordinates equal to 1e-12 relative (most are bit-identical; the
    decisions of the adaptive offsetter are the reference's, so a single flipped split would show as a count mismatch)."""
    import svgrasterize_amd as S

    z, meta = kat
    exact = total = 0
    for idx, m in enumerate(meta):
        path = S.Path.from_segments(z[f"{idx}_it"], z[f"{idx}_ip"], z[f"{idx}_is"])
        got = path.stroke(m["width"], m["cap"], m["join"])
        gt, gp, gs = [], [], []
        for sub in got.subpaths:
            gs.append(len(sub))
            for t, pts in sub:
                gt.append(t)
                q = np.zeros(8)
                q[: np.asarray(pts).size] = np.asarray(pts, dtype=np.float64).ravel()
                gp.append(q)
        want_t, want_p, want_s = z[f"{idx}_ot"], z[f"{idx}_op"], z[f"{idx}_os"]
        assert list(gs) == list(want_s), f"{m['name']}: subpath sizes {gs} != {list(want_s)}"
        assert list(gt) == list(want_t), f"{m['name']}: segment kinds differ"
        gp = np.array(gp).reshape(-1, 8)
        np.testing.assert_allclose(gp, want_p, rtol=1e-12, atol=1e-12, err_msg=m["name"])
        exact += int(np.sum(gp == want_p))
        total += gp.size
    assert exact / total > 0.99, f"only {exact}/{total} coordinates bit-identical"


def test_stroke_errors_and_empty():
    import svgrasterize_amd as S

    p = S.Path.from_svg("M0,0 L10,0 L10,10")
    with pytest.raises(ValueError):
        p.stroke(2.0, linecap="pointy")
    with pytest.raises(ValueError):
        p.stroke(2.0, linejoin="fancy")
    assert not S.Path([]).stroke(1.0)
    assert not S.Path.from_svg("M3,3 L3,3").stroke(1.0)  # nothing offsetable: no output (S:1146-1147)


@pytest.mark.gpu
def test_stroked_fill_matches_reference_render():
    """Scene.stroke end to end: tiger's stroke nodes stroked natively, then filled on the GPU, against the
    reference-stroked dump rendered the same way (identical outlines -> identical pixels)."""
    import svgrasterize_amd as S
    from svgrasterize_amd import Transform

    z = np.load(os.path.join(GOLD, "stroke_kat.npz"))
    meta = json.loads(str(z["meta"]))
    tr = Transform().matrix(0, 1, 0, 1, 0, 0).scale(0.5)
    checked = 0
    for idx, m in enumerate(meta):
        if not m["name"].startswith("tiger") or idx % 5:
            continue
        mine = S.Path.from_segments(z[f"{idx}_it"], z[f"{idx}_ip"], z[f"{idx}_is"]).stroke(m["width"], m["cap"], m["join"])
        ref = S.Path.from_segments(z[f"{idx}_ot"], z[f"{idx}_op"], z[f"{idx}_os"])
        a, b = mine.mask(tr), ref.mask(tr)
        assert (a is None) == (b is None)
        if a is None:
            continue
        assert tuple(a[0].offset) == tuple(b[0].offset) and a[0].image.shape == b[0].image.shape
        np.testing.assert_allclose(a[0].image, b[0].image, atol=1e-9)
        checked += 1
    assert checked >= 5


@pytest.mark.gpu
def test_scene_stroke_nodes_batched_and_per_node():
    """Scene.stroke leaves: through the batched route (solid paint: stroked once, then one batch with the fills) and
    through the per-node route (render of the node itself) -- both equal filling the reference-stroked outline."""
    import svgrasterize_amd as S
    from svgrasterize_amd import Transform

    z = np.load(os.path.join(GOLD, "stroke_kat.npz"))
    meta = json.loads(str(z["meta"]))
    tr = Transform().matrix(0, 1, 0, 1, 0, 0).scale(0.4)
    vp = [0, 0, 400, 400]
    rng = np.random.default_rng(7)
    mine, ref = [], []
    for idx, m in enumerate(meta):
        if not m["name"].startswith("tiger") or idx % 4:
            continue
        paint = np.concatenate([rng.uniform(0, 1, 3), [1.0]]) * rng.uniform(0.3, 1.0)
        src = S.Path.from_segments(z[f"{idx}_it"], z[f"{idx}_ip"], z[f"{idx}_is"])
        out = S.Path.from_segments(z[f"{idx}_ot"], z[f"{idx}_op"], z[f"{idx}_os"])
        mine.append(S.Scene.stroke(src, paint, m["width"], m["cap"], m["join"]))
        ref.append(S.Scene.fill(out, paint, None))
    a = S.Scene.group(mine).render(tr, viewport=vp, linear_rgb=False)
    b = S.Scene.group(ref).render(tr, viewport=vp, linear_rgb=False)
    assert tuple(a[0].offset) == tuple(b[0].offset) and a[0].image.shape == b[0].image.shape
    np.testing.assert_allclose(a[0].image, b[0].image, atol=1e-9)
    one = mine[0].render(tr, viewport=vp, linear_rgb=False)
    one_ref = ref[0].render(tr, viewport=vp, linear_rgb=False)
    np.testing.assert_allclose(one[0].image, one_ref[0].image, atol=1e-9)
    m1 = mine[0].render(tr, mask_only=True, viewport=vp)
    m1_ref = ref[0].render(tr, mask_only=True, viewport=vp)
    np.testing.assert_allclose(m1[0].image, m1_ref[0].image, atol=1e-9)


def test_degenerate_cubic_terminates():
    """A cubic whose first three points coincide asks to be split again and again without producing an offsetable piece;
    in the reference the halving only ends when rounding makes the piece collapse (S:2141-2145).  The native stroker
    drops a piece with no offsetable line at once: same outline, guaranteed termination, and no exception or
    allocation failure ever crosses the C ABI."""
    import svgrasterize_amd as S

    for d in ("M39.468,79.5577 C39.468,79.5577 39.468,79.5577 40.8281,77.7108 L10,10",
              "M5,5 C5,5 5,5 5,5 L9,9", "M1,1 C1,1 1,1 1,1 z", "M0,0 C0,0 1e-12,1e-12 3,4"):
        out = S.Path.from_svg(d).stroke(1.0, "round", "round")
        for sub in out.subpaths:
            for _kind, pts in sub:
                assert np.isfinite(np.asarray(pts, dtype=float)).all(), d
