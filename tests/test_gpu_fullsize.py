"""Full-size configurations of BASELINE.json on the GPU.

* tiger@2048 and material-design@4096: sparse pins of the reference's own full-size renders (about 16 k
  pixels each, biased to anti-aliased edges; oracle/gen_golden.py --full) -- float32 contract per pixel.
* synthetic 4096 paths @ 4096x4096 (the bench workload): size-independent properties -- the render is
  independent of how the canvas is sharded, idempotent, and a random subset of paths re-rendered alone
  through the CPU oracle on a cropped viewport matches the same crop of the GPU canvas.
"""
import json
import os

import numpy as np
import pytest

from tests.util import assert_f32_1ulp_rows, GOLDEN, assert_close64, assert_f32_1ulp

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def S():
    import svgrasterize_amd as S

    S.Context.get()
    return S


def _pins(name):
    from svgrasterize_amd import scenedump

    scene, info, z = scenedump.load_scene(os.path.join(GOLDEN, f"scene_{name}.npz"))
    return scene, info, z["full_idx"], z["full_val"]


def test_tiger_2048_batched(S):
    scene, info, idx, val = _pins("tiger")
    h, w = info["full"]["size"]
    tr = S.Transform().matrix(0, 1, 0, 1, 0, 0)
    img, st = S.render_canvas(scene, tr, [0, 0, h, w], linear_rgb=False)
    assert st.path_pixels == 34470049  # path-pixels of tiger@2048
    got = img.reshape(-1, 4)[idx]
    assert_f32_1ulp(got, val, what="tiger@2048 pins (batched)")
    # the reference CLI result is clipped to [0,1]; nothing else may be out of range
    assert img.min() >= 0.0 and img.max() <= 1.0


def test_tiger_2048_scene_render_matches_batched(S):
    scene, info, idx, val = _pins("tiger")
    h, w = info["full"]["size"]
    tr = S.Transform().matrix(0, 1, 0, 1, 0, 0)
    layer, _ = scene.render(tr, viewport=[0, 0, h, w], linear_rgb=False)
    assert [int(v) for v in layer.offset] == info["full"]["layer_offset"]
    assert list(layer.image.shape) == info["full"]["layer_shape"]
    canvas = layer.to_canvas_f32(h, w)
    assert_f32_1ulp(canvas.reshape(-1, 4)[idx], val, what="tiger@2048 pins (Scene.render)")


def test_material_4096_clips(S):
    """989 fills + 935 clip paths (per-node route: mask IN image) at the full 4096x4096."""
    scene, info, idx, val = _pins("material")
    h, w = info["full"]["size"]
    tr = S.Transform().matrix(0, 1, 0, 1, 0, 0)
    layer, _ = scene.render(tr, viewport=[0, 0, h, w], linear_rgb=False)
    assert [int(v) for v in layer.offset] == info["full"]["layer_offset"]
    assert list(layer.image.shape) == info["full"]["layer_shape"]
    canvas = layer.to_canvas_f32(h, w)
    assert_f32_1ulp(canvas.reshape(-1, 4)[idx], val, what="material@4096 pins")


def test_synth_4096_properties(S):
    from oracle import oracle as orc
    from svgrasterize_amd import _abi, dist as sdist, synth

    size, n = 4096, 4096
    sc = synth.make_scene(size, n)
    ctx = S.Context.get()
    batch = _abi.Batch(ctx, sc["segs"], sc["seg_kind"], sc["path_seg_off"], sc["path_m6"], sc["path_rule"], sc["path_paint"],
                       viewport=sc["viewport"])
    st = batch.plan()
    assert st.path_pixels == 160403147
    out = ctx.alloc(size * size * 16)
    batch.render(out, _abi.OUT_CANVAS_F32, _abi.RENDER_CLIP01)
    full = out.download((size, size, 4), np.float32)
    # idempotent: a second render of the same batch gives the same canvas up to LDS-atomic order (1 ULP ties)
    batch.render(out, _abi.OUT_CANVAS_F32, _abi.RENDER_CLIP01)
    again = out.download((size, size, 4), np.float32)
    # ... stated as the property it is: the few values that differ (order of the LDS float atomics inside a delta cell, decided at
    # a float32 rounding tie) differ by EXACTLY one float32 ULP -- neighbouring float32 numbers -- and never by more
    differs = full != again
    n_diff = int(differs.sum())
    assert n_diff <= full.size // 1_000_000, f"{n_diff} of {full.size} values differ between two renders"
    if n_diff:
        a, b = full[differs], again[differs]
        lo, hi = np.minimum(a, b), np.maximum(a, b)
        assert (np.nextafter(lo, np.float32(np.inf)) == hi).all(), "two renders differ by more than one float32 ULP somewhere"
    # sharding-invariant: rank 1 of 4 (strips of 16 bands) reproduces its rows of the full canvas
    tr = _abi.tile_rows()
    batch.set_bands(1, 4, 16)
    batch.plan()
    part = ctx.alloc(batch.owned_rows() * size * 16)
    batch.render(part, _abi.OUT_CANVAS_F32, _abi.RENDER_CLIP01)
    part = part.download((batch.owned_rows(), size, 4), np.float32)
    for k, (r0, r1) in enumerate(sdist.owned_row_ranges(size, tr, 1, 4, 16)):
        mine, theirs = part[k * tr: k * tr + (r1 - r0)], full[r0:r1]
        dd = mine != theirs
        assert dd.mean() < 1e-5
        if dd.any():   # (same property: neighbouring float32 numbers)
            lo, hi = np.minimum(mine[dd], theirs[dd]), np.maximum(mine[dd], theirs[dd])
            assert (np.nextafter(lo, np.float32(np.inf)) == hi).all()
    # the WHOLE canvas against the CPU oracle (64 row strips through the reference's own viewport mechanism, S:968-971, on the
    # host's cores): every one of the 67 M values of the bench scene inside the float32 contract
    ref, P, _ = orc.render_solid(synth.presentation_segs(sc), sc["seg_kind"], sc["path_seg_off"], sc["path_rule"],
                                 sc["path_paint"], sc["viewport"], clip01=True, strips=64, threads=orc.host_threads())
    assert_f32_1ulp_rows(full, ref, what="synth4096 whole canvas vs oracle")


def test_deterministic_render_is_bit_reproducible(S):
    """SVGR_RENDER_DETERMINISTIC: one wave per workgroup does every accumulation in list order, so two renders of the same
    batch are bit-identical (the reference's np.cumsum, S:983, is; the default render is only up to float32 rounding ties).
    Same pixels as the default render within the contract, and across a re-plan."""
    from oracle import oracle as orc
    from svgrasterize_amd import _abi, synth

    size, n = 2048, 1500
    sc = synth.make_scene(size, n)
    ctx = S.Context.get()
    flags = _abi.RENDER_CLIP01 | _abi.RENDER_DETERMINISTIC
    canv = []
    for _ in range(2):  # (two batches: a new plan, new buffers)
        batch = _abi.Batch(ctx, sc["segs"], sc["seg_kind"], sc["path_seg_off"], sc["path_m6"], sc["path_rule"], sc["path_paint"],
                           viewport=sc["viewport"])
        batch.plan()
        out = ctx.alloc(size * size * 32)
        for _k in range(3):
            batch.render(out, _abi.OUT_CANVAS_F64, flags)
            canv.append(out.download((size, size, 4), np.float64))
        batch.destroy()
    for c in canv[1:]:
        assert np.array_equal(c, canv[0]), "deterministic renders differ"
    ref, _, _ = orc.render_solid(synth.presentation_segs(sc), sc["seg_kind"], sc["path_seg_off"], sc["path_rule"],
                                 sc["path_paint"], sc["viewport"], clip01=True, strips=32, threads=orc.host_threads())
    assert_close64(canv[0], ref, atol=1e-10, what="deterministic render vs oracle")


def test_planned_slab_places_do_not_change_the_picture(S, monkeypatch):
    """The staged plan hands every path its slabs' places in k_path_build's work list (heaviest first: host arithmetic that
    must mirror k_path_bbox's).  With and without the plan's places (SVGR_SAFE_PATH: every place from the device's cursors) a deterministic
    render gives the same bits -- on the whole drawing and on one rank's bands of a two-rank sharding (slabs without an
    owned band are left out of both counts)."""
    from svgrasterize_amd import _abi, synth

    size, n = 2048, 1500
    sc = synth.make_scene(size, n)
    ctx = S.Context.get()
    flags = _abi.RENDER_CLIP01 | _abi.RENDER_DETERMINISTIC

    def render(bands):
        batch = _abi.Batch(ctx, sc["segs"], sc["seg_kind"], sc["path_seg_off"], sc["path_m6"], sc["path_rule"], sc["path_paint"],
                           viewport=sc["viewport"])
        if bands:
            batch.set_bands(*bands)
        batch.plan()
        rows = batch.owned_rows() if bands else size
        out = ctx.alloc(max(rows, 1) * size * 32)
        batch.render(out, _abi.OUT_CANVAS_F64, flags)
        got = out.download((rows, size, 4), np.float64)
        batch.destroy()
        return got

    for bands in (None, (1, 2, 16)):
        monkeypatch.delenv("SVGR_SAFE_PATH", raising=False)
        a = render(bands)
        assert np.abs(a).max() > 0
        # every place a planned render takes from its plan -- the slabs' (heaviest first), the band lists' and the lists themselves
        # (k_band_entries not launched), the lanes' edge places in k_flatten, the cells' add places (k_path_build<true>: one pass) --
        # against the same render with all of them found again by the device's cursors and a second pass (SVGR_SAFE_PATH, read when
        # the plan is made): the same lists, the same bits
        monkeypatch.setenv("SVGR_SAFE_PATH", "1")
        b = render(bands)
        monkeypatch.delenv("SVGR_SAFE_PATH", raising=False)
        assert np.array_equal(a, b), f"the plan's places change the picture (bands {bands})"


def test_synth_8192_config4_windows(S):
    """BASELINE config 4 (10 000 random cubic paths @ 8192x8192) on one GPU: more paths than one band-list pass keeps in
    registers, 6 M record slots, a 1 GiB canvas.  Windows of the canvas against the CPU oracle rendered through the
    reference's own viewport mechanism, plus one rank's strips of an 8-way sharding against the same rows."""
    from oracle import oracle as orc
    from svgrasterize_amd import _abi, dist as sdist, synth

    size, n = 8192, 10000
    sc = synth.make_scene(size, n)
    ctx = S.Context.get()
    batch = _abi.Batch(ctx, sc["segs"], sc["seg_kind"], sc["path_seg_off"], sc["path_m6"], sc["path_rule"], sc["path_paint"],
                       viewport=sc["viewport"])
    st = batch.plan()
    assert st.path_pixels == 399658495 and st.n_nonempty == n
    out = ctx.alloc(size * size * 16)
    batch.render(out, _abi.OUT_CANVAS_F32, _abi.RENDER_CLIP01)
    full = out.download((size, size, 4), np.float32)
    # the WHOLE canvas (268 M values) against the CPU oracle, 128 row strips on the host's cores
    ref, _, _ = orc.render_solid(synth.presentation_segs(sc), sc["seg_kind"], sc["path_seg_off"], sc["path_rule"],
                                 sc["path_paint"], sc["viewport"], clip01=True, strips=128, threads=orc.host_threads())
    assert_f32_1ulp_rows(full, ref, what="synth8192 whole canvas vs oracle")
    del ref
    tr = _abi.tile_rows()
    strip = max(1, 128 // tr)
    batch.set_bands(5, 8, strip)
    batch.plan()
    part = ctx.alloc(batch.owned_rows() * size * 16)
    batch.render(part, _abi.OUT_CANVAS_F32, _abi.RENDER_CLIP01)
    part = part.download((batch.owned_rows(), size, 4), np.float32)
    k = 0
    for r0, r1 in sdist.owned_row_ranges(size, tr, 5, 8, strip):
        d = np.abs(part[k: k + (r1 - r0)].astype(np.float64) - full[r0:r1])
        assert d.max() < 2e-7 and (d > 0).mean() < 1e-5
        k += r1 - r0


@pytest.mark.gpu
def test_largest_canvas_windows_equal_small_viewport_renders():
    """A 16384 x 16384 canvas (the planner's tiles x bands at their largest in the configs' spirit): shapes in the four
    corners and the middle, checked through windows of the big render against renders of the same scene with the window
    as its viewport (the reference's viewport cropping, S:966-975, is translation of the same pixels)."""
    import svgrasterize_amd as S
    from svgrasterize_amd import _abi
    from svgrasterize_amd.scene import build_batch

    ctx = S.Context.get()
    size = 16384
    swap = S.Transform().matrix(0, 1, 0, 1, 0, 0)
    rng = np.random.default_rng(99)
    leaves, spots = [], [(40, 50), (40, size - 300), (size - 280, 30), (size - 260, size - 270), (size // 2 - 100, size // 2 - 90)]
    for k, (r0, c0) in enumerate(spots):
        for j in range(6):
            cx, cy = c0 + 120 + 30 * rng.uniform(-1, 1), r0 + 110 + 30 * rng.uniform(-1, 1)
            rad = rng.uniform(40, 110)
            d = (f"M{cx - rad},{cy} C{cx - rad},{cy - rad * 1.3} {cx + rad * 0.4},{cy - rad} {cx + rad},{cy - 0.2 * rad} "
                 f"S{cx + 0.3 * rad},{cy + rad * 1.2} {cx - rad},{cy} Z")
            a = rng.uniform(0.3, 1.0)
            paint = np.array([*(rng.uniform(0, 1, 3) * a), a])
            leaves.append((S.Path.from_svg(d), swap.m6(), 1 if j % 3 == 2 else 0, paint, 0))  # (rule 1 = evenodd)
    # one long diagonal sliver crossing every band and column tile
    leaves.append((S.Path.from_svg(f"M3,5 L{size - 4},{size - 9} L{size - 9},{size - 3} Z"), swap.m6(), 0, np.array([0.1, 0.2, 0.3, 0.5]), 0))

    def render(viewport):
        batch = build_batch(leaves, viewport, ctx)
        batch.plan()
        out = ctx.alloc(viewport[2] * viewport[3] * 16)
        batch.render(out, _abi.OUT_CANVAS_F32, _abi.RENDER_CLIP01)
        return batch, out

    big_batch, big = render([0, 0, size, size])
    assert big_batch.stats.n_edges > 0
    for r0, c0 in spots + [(8000, 8000)]:
        r0, c0 = min(max(r0 - 20, 0), size - 320), min(max(c0 - 20, 0), size - 320)
        _small_batch, small = render([r0, c0, 320, 320])
        want = small.download((320, 320, 4), np.float32)
        got = np.stack([big.download((320, 4), np.float32, offset=((r0 + i) * size + c0) * 16) for i in range(320)])
        # (same pixels up to the order of the double sums: what lies left of a window is folded into its first column)
        assert_f32_1ulp(got, want.astype(np.float64), what=f"window at ({r0}, {c0})")
        assert want[..., 3].max() > 0.05  # the window really shows something (the sliver at least)
