"""Randomised parity: seeded synthetic scenes of random size / path count / viewport (also partly or wholly off the
drawing) on the GPU against the CPU oracle, both planners, f64 and f32 outputs, and the multi-mask output."""
import numpy as np
import pytest

from tests.util import assert_close64, assert_f32_1ulp

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def S():
    import svgrasterize_amd as S

    S.Context.get()
    return S


@pytest.mark.parametrize("seed", list(range(14)))
def test_random_scene_vs_oracle(S, seed):
    from oracle import oracle as orc
    from svgrasterize_amd import _abi, synth

    rng = np.random.default_rng(1000 + seed)
    size = int(rng.integers(40, 900))
    n = int(rng.integers(1, 160)) if seed % 3 else int(rng.integers(1, 6))  # small batches take the one-pass planner
    sc = synth.make_scene(size, n, seed=synth.SEED + 17 * seed)
    # a viewport somewhere around the drawing: inside, overlapping an edge, or far away (nothing to draw)
    rows, cols = int(rng.integers(8, size + 60)), int(rng.integers(8, size + 60))
    r0, c0 = int(rng.integers(-80, size)), int(rng.integers(-80, size))
    if seed == 5:
        r0, c0 = size + 500, size + 500
    vp = (r0, c0, rows, cols)
    ref, P, _E = orc.render_solid(synth.presentation_segs(sc), sc["seg_kind"], sc["path_seg_off"], sc["path_rule"],
                                  sc["path_paint"], vp, clip01=True)
    ctx = S.Context.get()
    batch = _abi.Batch(ctx, sc["segs"], sc["seg_kind"], sc["path_seg_off"], sc["path_m6"], sc["path_rule"], sc["path_paint"],
                       viewport=vp)
    st = batch.plan()
    assert st.path_pixels == P
    out64 = ctx.alloc(rows * cols * 32)
    batch.render(out64, _abi.OUT_CANVAS_F64, _abi.RENDER_CLIP01)
    assert_close64(out64.download((rows, cols, 4), np.float64), ref, atol=1e-10, what=f"seed {seed} f64")
    out32 = ctx.alloc(rows * cols * 16)
    for _ in range(2):  # (the second render recomputes the geometry instead of reusing the plan's pass)
        batch.render(out32, _abi.OUT_CANVAS_F32, _abi.RENDER_CLIP01)
        assert_f32_1ulp(out32.download((rows, cols, 4), np.float32), ref, what=f"seed {seed} f32")
    # every path's mask in one launch == the oracle's per-path masks
    buf, offs, bb = batch.render_masks()
    flat = buf.download((max(int(offs[-1]), 1),), np.float64)
    pres = synth.presentation_segs(sc)
    for p in range(0, n, max(1, n // 7)):
        s0, s1 = int(sc["path_seg_off"][p]), int(sc["path_seg_off"][p + 1])
        res = orc.path_mask(np.zeros((0, 2, 2)), pres[s0:s1].reshape(-1, 4, 2), None, "evenodd" if sc["path_rule"][p] else None, vp)
        if res is None:
            assert bb[p, 2] <= 0 or bb[p, 3] <= 0
            continue
        mask, off, _edges = res
        assert (int(bb[p, 0]), int(bb[p, 1])) == (int(off[0]), int(off[1])) and tuple(mask.shape[:2]) == (int(bb[p, 2]), int(bb[p, 3]))
        got = flat[int(offs[p]): int(offs[p + 1])].reshape(int(bb[p, 2]), int(bb[p, 3]))
        assert_close64(got, np.asarray(mask).reshape(got.shape), atol=1e-11, what=f"seed {seed} mask {p}")


@pytest.mark.parametrize("seed", list(range(6)))
def test_random_affine_per_path_vs_oracle(S, seed):
    """Every path under its own random affine map (rotation, skew, non-uniform scale, translation) applied on the device
    (fma form of S:531-534), against the oracle fed with the same points transformed by numpy on the host."""
    from oracle import oracle as orc
    from svgrasterize_amd import _abi, synth

    rng = np.random.default_rng(5000 + seed)
    size = int(rng.integers(60, 600))
    n = int(rng.integers(1, 80))
    sc = synth.make_scene(size, n, seed=synth.SEED + 101 * seed)
    m6 = np.zeros((n, 6))
    pres = np.zeros_like(sc["segs"])
    for p in range(n):
        ang, shear = rng.uniform(0, 2 * np.pi), rng.uniform(-0.6, 0.6)
        sx, sy = rng.uniform(0.4, 1.8, 2)
        lin = np.array([[np.cos(ang), -np.sin(ang)], [np.sin(ang), np.cos(ang)]]) @ np.array([[1, shear], [0, 1]]) @ np.diag([sx, sy])
        c = np.array([size / 2, size / 2])
        t = c - lin @ c + rng.uniform(-0.2, 0.2, 2) * size
        user = np.array([[lin[0, 0], lin[0, 1], t[0]], [lin[1, 0], lin[1, 1], t[1]], [0, 0, 1]])
        tr = S.Transform().matrix(0, 1, 0, 1, 0, 0) @ S.Transform(user)  # presentation space = swap . user map
        m6[p] = tr.m6()
        s0, s1 = int(sc["path_seg_off"][p]), int(sc["path_seg_off"][p + 1])
        pres[s0:s1] = tr(sc["segs"][s0:s1].reshape(-1, 2)).reshape(-1, 8)
    rows, cols = int(rng.integers(16, size + 40)), int(rng.integers(16, size + 40))
    vp = (int(rng.integers(-40, size // 2)), int(rng.integers(-40, size // 2)), rows, cols)
    ref, P, _E = orc.render_solid(pres, sc["seg_kind"], sc["path_seg_off"], sc["path_rule"], sc["path_paint"], vp, clip01=True)
    ctx = S.Context.get()
    batch = _abi.Batch(ctx, sc["segs"], sc["seg_kind"], sc["path_seg_off"], m6, sc["path_rule"], sc["path_paint"], viewport=vp)
    st = batch.plan()
    assert st.path_pixels == P
    out = ctx.alloc(rows * cols * 32)
    batch.render(out, _abi.OUT_CANVAS_F64, _abi.RENDER_CLIP01)
    assert_close64(out.download((rows, cols, 4), np.float64), ref, atol=1e-10, what=f"affine seed {seed}")
