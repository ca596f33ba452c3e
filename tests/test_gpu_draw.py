"""svgr_batch_draw (ABI 6): plan + render behind ONE wait -- a frame with new geometry, the reference's only mode (every
`Path.mask` flattens and rasterises from scratch, S:948-957) -- against svgr_batch_plan + svgr_batch_render of the same inputs and
against the CPU oracle.  A batch planned before takes the single pass with ONE flatten traversal (k_flatten<.., SCAN>: count,
decoupled look-back, store); a new batch the two-pass plan with the tile kernel behind its second pass; a guess that does not hold
must cost the staged plan, never a fault or a wrong picture."""
import numpy as np
import pytest

from tests.util import assert_close64, assert_f32_1ulp, sort_edges

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def S():
    import svgrasterize_amd as S

    S.Context.get()
    return S


def _new(S, sc, m6=None):
    from svgrasterize_amd import _abi

    return _abi.Batch(S.Context.get(), sc["segs"], sc["seg_kind"], sc["path_seg_off"], sc["path_m6"] if m6 is None else m6,
                      sc["path_rule"], sc["path_paint"], viewport=list(sc["viewport"]))


def _moved(sc, k):
    m6 = np.array(sc["path_m6"], dtype=np.float64, copy=True)
    m6[:, 2] += 0.125 * k + 3.0 * (k % 2)    # (rows: a fraction of a pixel, and every other frame three whole ones)
    m6[:, 5] += 0.0625 * k
    return m6


def _oracle(sc, m6, clip01=False):
    from oracle import oracle as orc

    pts = sc["segs"].reshape(-1, 4, 2)
    pres = np.empty_like(pts)
    seg_path = np.repeat(np.arange(len(sc["path_seg_off"]) - 1), np.diff(sc["path_seg_off"]))
    for p in np.unique(seg_path):
        m = np.eye(3)
        m[:2, :] = m6[p].reshape(2, 3)
        pres[seg_path == p] = orc.transform_points(m, pts[seg_path == p])
    ref, _, _ = orc.render_solid(pres.reshape(-1, 8), sc["seg_kind"], sc["path_seg_off"], sc["path_rule"], sc["path_paint"],
                                 sc["viewport"], clip01=clip01)
    return ref


@pytest.mark.parametrize("size,n", [(1024, 1000), (1024, 300), (320, 12)])   # (two-pass plan / size models; the last: a launch that does not fill the chip, 64 lanes per segment)
def test_draw_is_plan_plus_render(S, size, n):
    """A new batch drawn in one call, then moved and drawn again (five frames: the re-plan's single pass), then replayed: every
    canvas equals the one svgr_batch_plan + svgr_batch_render give for the same transforms (1e-12: the order of the LDS atomics) and
    the CPU oracle's (1e-10); the edge arrays -- made by ONE flatten traversal in the re-plans -- are the staged plan's, bit for
    bit and in the same order; the statistics and bboxes a draw leaves are the plan's."""
    from svgrasterize_amd import _abi, synth

    sc = synth.make_scene(size, n)
    ctx = S.Context.get()
    out = ctx.alloc(size * size * 32)
    b = _new(S, sc)
    b.draw(out, _abi.OUT_CANVAS_F64)                       # cold: the two-pass plan (or the small batches' single pass) + tile kernel
    got = out.download((size, size, 4), np.float64)
    assert_close64(got, _oracle(sc, np.asarray(sc["path_m6"], dtype=np.float64)), atol=1e-10, what="cold draw vs oracle")
    for k in range(1, 6):
        m6 = _moved(sc, k)
        b.set_transforms(m6)
        b.draw(out, _abi.OUT_CANVAS_F64)                   # re-plan: ONE flatten traversal, everything behind one wait
        got = out.download((size, size, 4), np.float64)
        ref_b = _new(S, sc, m6)
        st_ref = ref_b.plan()
        ref_out = ctx.alloc(size * size * 32)
        ref_b.render(ref_out, _abi.OUT_CANVAS_F64)
        want = ref_out.download((size, size, 4), np.float64)
        assert want.any()
        assert_close64(got, want, atol=1e-12, what=f"draw after set_transforms, frame {k}")
        st = b.stats
        assert (st.n_edges, st.path_pixels, st.n_path_bands) == (st_ref.n_edges, st_ref.path_pixels, st_ref.n_path_bands)
        assert np.array_equal(b.bboxes(), ref_b.bboxes())
        e, ep = b.edges()
        e_ref, ep_ref = ref_b.edges()
        assert np.array_equal(e, e_ref) and np.array_equal(ep, ep_ref), "the one-traversal flatten stored other edges, or elsewhere"
        if k == 5:
            assert_close64(got, _oracle(sc, m6), atol=1e-10, what="re-planned draw vs oracle")
            # ... and the plan a draw leaves replays like any other (the first replay makes the slab order)
            for _ in range(2):
                b.render(out, _abi.OUT_CANVAS_F64)
                assert_close64(out.download((size, size, 4), np.float64), want, atol=1e-12, what="replay of a draw's plan")
            b.draw(out, _abi.OUT_CANVAS_F64)               # (planned already: a render + a wait)
            assert_close64(out.download((size, size, 4), np.float64), want, atol=1e-12, what="draw of a planned batch")
        ref_b.destroy()
    b.destroy()


def test_draw_float32_canvas_is_inside_the_contract(S):
    """The production output (float32 canvas, clip01) of a cold draw and of a re-planned draw against the oracle under the
    float32 contract: whole canvas."""
    from svgrasterize_amd import _abi, synth

    size, n = 1536, 1200
    sc = synth.make_scene(size, n)
    ctx = S.Context.get()
    out = ctx.alloc(size * size * 16)
    b = _new(S, sc)
    b.draw(out, _abi.OUT_CANVAS_F32, _abi.RENDER_CLIP01)
    assert_f32_1ulp(out.download((size, size, 4), np.float32), _oracle(sc, np.asarray(sc["path_m6"], dtype=np.float64), clip01=True), what="cold draw f32")
    m6 = _moved(sc, 3)
    b.set_transforms(m6)
    b.draw(out, _abi.OUT_CANVAS_F32, _abi.RENDER_CLIP01)
    assert_f32_1ulp(out.download((size, size, 4), np.float32), _oracle(sc, m6, clip01=True), what="re-planned draw f32")
    b.destroy()


def test_draw_falls_back_when_a_capacity_does_not_hold(S, monkeypatch):
    """The tile kernel runs BEHIND a pass whose capacities are guesses.  (a) A drawing that grows under the buffers of its last plan
    (set_transforms scales it up three times): the single pass flags the overflow, writes nothing outside its arrays, the tile
    kernel behind it reads only what was written, and the staged plan + an ordinary render give the right picture.  (b) A new
    batch whose add-list guess is too small on purpose (SVGR_TWO_PASS_SHRINK): the same."""
    from oracle import oracle as orc
    from svgrasterize_amd import _abi, synth

    ctx = S.Context.get()
    swap = S.Transform().matrix(0, 1, 0, 1, 0, 0)
    path = S.Path.from_svg("M20,10 L60,300 L100,30 L140,290 L180,15 C300,70 260,230 200,295 L20,290 Z")
    segs, kinds = path.packed()
    vp = [0, 0, 960, 960]
    paint = np.array([[0.2, 0.3, 0.1, 0.5]])
    b = _abi.Batch(ctx, segs, kinds, [0, len(segs)], swap.m6(), [0], paint, viewport=vp)
    canvas = ctx.alloc(vp[2] * vp[3] * 32)
    b.draw(canvas, _abi.OUT_CANVAS_F64)
    small = canvas.download((vp[2], vp[3], 4), np.float64)
    assert small.any()
    b.set_transforms(swap.scale(3.0).m6())
    b.draw(canvas, _abi.OUT_CANVAS_F64)
    got = canvas.download((vp[2], vp[3], 4), np.float64)
    m3 = np.array([[0.0, 3.0, 0.0], [3.0, 0.0, 0.0], [0.0, 0.0, 1.0]])
    want, _, _ = orc.render_solid(orc.transform_points(m3, segs.reshape(-1, 4, 2)).reshape(-1, 8), kinds, [0, len(segs)], [0], paint, vp, clip01=False)
    assert_close64(got, want, atol=1e-10, what="draw of a drawing that outgrew its buffers")
    b.destroy()
    # (b)
    size, n = 1024, 1000
    sc = synth.make_scene(size, n)
    ref_b = _new(S, sc)
    ref_b.plan()
    out = ctx.alloc(size * size * 32)
    ref_b.render(out, _abi.OUT_CANVAS_F64)
    want = out.download((size, size, 4), np.float64)
    ref_b.destroy()
    monkeypatch.setenv("SVGR_TWO_PASS_SHRINK", "6")
    b = _new(S, sc)
    b.draw(out, _abi.OUT_CANVAS_F64)
    monkeypatch.delenv("SVGR_TWO_PASS_SHRINK")
    assert_close64(out.download((size, size, 4), np.float64), want, atol=1e-12, what="draw with an add-list guess that was too small")
    b.destroy()


def test_frames_from_new_batches_inherit_capacities_not_contents(S, monkeypatch):
    """A caller that makes a new batch per frame: a destroyed large batch leaves its work arrays to its context and the next batch
    for the same viewport plans in ONE pass on the inherited capacities.  Nothing of the old batch may show: frames of a moving
    drawing, then a DIFFERENT drawing (another seed, 1.4 times the paths: some capacity overflows and the two-pass plan takes over),
    then a much smaller one (does not adopt), every one against a batch planned without inheritance."""
    from svgrasterize_amd import _abi, synth

    size = 1024
    ctx = S.Context.get()
    out = ctx.alloc(size * size * 32)

    def reference(sc, m6):
        monkeypatch.setenv("SVGR_NO_SPARE", "1")
        b = _new(S, sc, m6)
        st = b.plan()
        ref_out = ctx.alloc(size * size * 32)
        b.render(ref_out, _abi.OUT_CANVAS_F64)
        img = ref_out.download((size, size, 4), np.float64)
        bb, edges = b.bboxes().copy(), b.edges()
        b.destroy()
        monkeypatch.delenv("SVGR_NO_SPARE")
        return st, img, bb, edges

    scenes = [synth.make_scene(size, 1000)] * 4 + [synth.make_scene(size, 1400), synth.make_scene(size, 1000), synth.make_scene(size, 100)]
    prev = None
    for k, sc in enumerate(scenes):
        m6 = _moved(sc, k)
        b = _new(S, sc, m6)
        if prev is not None:
            prev.destroy()          # (the frame before leaves its arrays behind)
        if k % 2:
            b.draw(out, _abi.OUT_CANVAS_F64)
        else:
            b.plan()
            b.render(out, _abi.OUT_CANVAS_F64)
        got = out.download((size, size, 4), np.float64)
        st_ref, want, bb_ref, (e_ref, ep_ref) = reference(sc, m6)
        assert want.any()
        assert_close64(got, want, atol=1e-12, what=f"frame {k} from a new batch")
        st = b.stats
        assert (st.n_edges, st.path_pixels) == (st_ref.n_edges, st_ref.path_pixels)
        assert np.array_equal(b.bboxes(), bb_ref)
        e, ep = b.edges()
        assert np.array_equal(e, e_ref) and np.array_equal(ep, ep_ref)
        b.render(out, _abi.OUT_CANVAS_F64)      # ... and replays like any other plan
        assert_close64(out.download((size, size, 4), np.float64), want, atol=1e-12, what=f"replay of frame {k}")
        prev = b
    prev.destroy()


def test_measure_helpers_report_device_time_and_launches(S):
    """svgr_measure_begin / _end: the device time of what is enqueued between them, behind a hold of the stream (so that the host's
    pace does not show); svgr_measure_launches: the library's own count of kernel launches.  A planned render of the bench's kind is
    five launches; its device time is a fraction of a millisecond and does not include the hold."""
    from svgrasterize_amd import _abi, synth

    size = 1024
    sc = synth.make_scene(size, 1000)
    ctx = S.Context.get()
    b = _new(S, sc)
    out = ctx.alloc(size * size * 16)
    b.draw(out, _abi.OUT_CANVAS_F32, _abi.RENDER_CLIP01)
    b.render(out, _abi.OUT_CANVAS_F32, _abi.RENDER_CLIP01)     # (the first replay: the slab order)
    ctx.sync()
    n0 = ctx.launches()
    ctx.measure_begin(3.0)
    b.render(out, _abi.OUT_CANVAS_F32, _abi.RENDER_CLIP01)
    ms = ctx.measure_end()
    assert ctx.launches() - n0 == 1 + 5, "the hold + flatten, bbox, path build, tile lists, tile kernel"
    assert 0.005 < ms < 1.0, ms
    ctx.measure_begin(0.0)
    assert ctx.measure_end() < 0.5
    b.destroy()


def test_draw_with_groups_clips_bands_and_without_a_viewport(S, monkeypatch):
    """svgr_batch_draw off the production variant: a document's batch with clip pairs and isolated groups (material-design: the
    clip / group variants of the tile kernel behind the unvalidated pass), moved and drawn again; a batch WITHOUT a viewport and one
    restricted to a rank's bands (no fast path: plan, render, wait); SVGR_RENDER_DETERMINISTIC (likewise).  Each against
    svgr_batch_plan + svgr_batch_render of the same inputs."""
    import os

    from svgrasterize_amd import _abi, scenedump, synth
    from svgrasterize_amd import scene as sm

    ctx = S.Context.get()
    scene, info, _z = scenedump.load_scene(os.path.join(os.path.dirname(__file__), "golden", "scene_material.npz"))
    size = 1024
    tr = S.Transform().matrix(0, 1, 0, 1, 0, 0).scale(size / info["full"]["size"][0])
    leaves = sm._drop_empty(sm._batchable_leaves(scene, tr, False))
    vp = [0, 0, size, size]
    out = ctx.alloc(size * size * 32)
    ref_out = ctx.alloc(size * size * 32)
    b = sm.build_batch(leaves, vp)
    ref = sm.build_batch(leaves, vp)
    ref.plan()
    ref.render(ref_out, _abi.OUT_CANVAS_F64)
    want = ref_out.download((size, size, 4), np.float64)
    b.draw(out, _abi.OUT_CANVAS_F64)
    assert want.any()
    assert_close64(out.download((size, size, 4), np.float64), want, atol=1e-12, what="draw of a batch with clips and groups")
    m6 = np.array([lf[1] for lf in leaves], dtype=np.float64).reshape(len(leaves), 6).copy()
    m6[:, 2] += 7.25
    m6[:, 5] -= 3.5
    for bb in (b, ref):
        bb.set_transforms(m6)
    ref.plan()
    ref.render(ref_out, _abi.OUT_CANVAS_F64)
    want = ref_out.download((size, size, 4), np.float64)
    b.draw(out, _abi.OUT_CANVAS_F64)
    assert_close64(out.download((size, size, 4), np.float64), want, atol=1e-12, what="re-planned draw of a batch with clips and groups")
    assert np.array_equal(b.bboxes(), ref.bboxes())
    b.destroy(); ref.destroy()
    # no viewport: the union of the bboxes is the canvas (S:968); a rank's bands; a deterministic draw
    sc = synth.make_scene(512, 160)
    for mode in ("no viewport", "bands", "deterministic"):
        def make():
            return _abi.Batch(ctx, sc["segs"], sc["seg_kind"], sc["path_seg_off"], sc["path_m6"], sc["path_rule"], sc["path_paint"],
                              viewport=None if mode == "no viewport" else list(sc["viewport"]))
        a, r = make(), make()
        flags = _abi.RENDER_DETERMINISTIC if mode == "deterministic" else 0
        if mode == "bands":
            a.set_bands(1, 2, 4); r.set_bands(1, 2, 4)
        st = r.plan()
        rows = int(r.owned_rows()) if mode == "bands" else int(st.bbox_union[2]) if mode == "no viewport" else 512
        cols = int(st.bbox_union[3]) if mode == "no viewport" else 512
        o1, o2 = ctx.alloc(max(rows * cols * 32, 32)), ctx.alloc(max(rows * cols * 32, 32))
        r.render(o2, _abi.OUT_CANVAS_F64, flags)
        a.draw(o1, _abi.OUT_CANVAS_F64, flags)
        got, want = o1.download((rows, cols, 4), np.float64), o2.download((rows, cols, 4), np.float64)
        assert want.any(), mode
        if mode == "deterministic":
            assert np.array_equal(got, want), mode
        else:
            assert_close64(got, want, atol=1e-12, what=f"draw, {mode}")
        a.destroy(); r.destroy()
