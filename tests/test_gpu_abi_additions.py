"""The C-ABI entry points round 4 added, each against the entry point it stands in for (same library, same inputs):
SVGR_RENDER_SAME_GEOMETRY, svgr_batch_render_windows, SVGR_OUT_FILLS_F64, svgr_layer_convert_to / svgr_layer_scale_to."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def S():
    import svgrasterize_amd as S

    S.Context.get()
    return S


def _batch(S, size=512, n=160):
    from svgrasterize_amd import _abi, synth

    sc = synth.make_scene(size, n)
    ctx = S.Context.get()
    batch = _abi.Batch(ctx, sc["segs"], sc["seg_kind"], sc["path_seg_off"], sc["path_m6"], sc["path_rule"], sc["path_paint"],
                       viewport=list(sc["viewport"]))
    batch.plan()
    return ctx, batch, sc


def _window(ctx, batch, win, flags=0):
    from svgrasterize_amd import _abi

    out = ctx.alloc(win[2] * win[3] * 32)
    batch.render(out, _abi.OUT_CANVAS_F64, flags, window=win)
    return out.download((win[2], win[3], 4), np.float64)


def test_same_geometry_windows_are_the_windows(S):
    """A window drawn with SVGR_RENDER_SAME_GEOMETRY behind another window of the same batch is the window a render with its
    own geometry pass draws (to the order of the LDS atomics), and equals that part of the whole canvas; after a setter the
    flag is ignored: the moved picture is drawn, not the old one."""
    from svgrasterize_amd import _abi

    ctx, batch, sc = _batch(S)
    size = sc["viewport"][2]
    whole = ctx.alloc(size * size * 32)
    batch.render(whole, _abi.OUT_CANVAS_F64)
    canvas = whole.download((size, size, 4), np.float64)
    wins = [(16, 32, 200, 300), (250, 0, 262, 512), (3, 401, 77, 90)]
    first = _window(ctx, batch, wins[0])
    assert np.abs(first - canvas[16:216, 32:332]).max() <= 1e-12
    for w in wins[1:]:
        same = _window(ctx, batch, w, _abi.RENDER_SAME_GEOMETRY)
        assert np.abs(same - canvas[w[0]:w[0] + w[2], w[1]:w[1] + w[3]]).max() <= 1e-12
        assert same.any()
    # a setter in between: the geometry is produced again although the flag is given
    m6 = np.array(sc["path_m6"], dtype=np.float64)
    m6[:, 2] += 40.0   # (40 rows down)
    batch.set_transforms(m6)
    batch.plan()
    _window(ctx, batch, wins[0])
    batch.set_transforms(np.array(sc["path_m6"], dtype=np.float64))
    batch.plan()
    back = _window(ctx, batch, wins[1], _abi.RENDER_SAME_GEOMETRY)   # (first render after the plan: the plan's own pass)
    assert np.abs(back - canvas[250:512, 0:512]).max() <= 1e-12
    batch.destroy()


def test_windows_side_by_side_are_the_windows_one_by_one(S):
    from svgrasterize_amd import _abi

    ctx, batch, sc = _batch(S)
    wins = [(0, 0, 128, 128), (100, 200, 300, 150), (17, 33, 45, 67), (256, 256, 256, 256), (0, 448, 512, 64),
            (300, 0, 100, 512), (5, 5, 500, 500), (64, 64, 64, 64), (400, 100, 90, 333), (128, 0, 16, 512)]
    one_by_one = [_window(ctx, batch, w) for w in wins]
    outs = [ctx.alloc(w[2] * w[3] * 32) for w in wins]
    for _ in range(2):   # (twice: the side streams are reused, the outputs overwritten)
        batch.render_windows(outs, _abi.OUT_CANVAS_F64, wins)
        for w, o, ref in zip(wins, outs, one_by_one):
            got = o.download((w[2], w[3], 4), np.float64)
            assert np.abs(got - ref).max() <= 1e-12
    # more windows than one launch's table holds (64): several launches, the same layers
    many = [(r, c, 40, 50) for r in range(0, 480, 48) for c in range(0, 480, 60)][:70]
    m_outs = [ctx.alloc(40 * 50 * 32) for _ in many]
    batch.render_windows(m_outs, _abi.OUT_CANVAS_F64, many)
    full = _window(ctx, batch, (0, 0, 512, 512))
    for w, o in zip(many, m_outs):
        assert np.abs(o.download((40, 50, 4), np.float64) - full[w[0]:w[0] + 40, w[1]:w[1] + 50]).max() <= 1e-12
    # behind the call the context's stream is behind the windows: an operation enqueued now sees them
    S.Layer._from_device(outs[3], (256, 256, 4), (0, 0), True, False).opacity(0.5)
    with pytest.raises(ValueError):
        batch.render_windows(outs[:1], _abi.OUT_CANVAS_F64, [(0, 0, 600, 10)])   # not inside the viewport
    batch.destroy()


def test_fill_layers_of_all_paths_are_the_single_path_fills(S):
    from svgrasterize_amd import _abi, synth

    ctx, batch, sc = _batch(S, 384, 24)
    buf, offs, bb = batch.render_fills()
    assert int(offs[-1]) > 0
    for p in range(24):
        rows, cols = int(bb[p, 2]), int(bb[p, 3])
        if rows <= 0 or cols <= 0:
            continue
        got = buf.download((rows, cols, 4), np.float64, offset=int(offs[p]) * 32)
        s0, s1 = int(sc["path_seg_off"][p]), int(sc["path_seg_off"][p + 1])
        one = _abi.Batch(ctx, sc["segs"][s0:s1], sc["seg_kind"][s0:s1], [0, s1 - s0], sc["path_m6"][p:p + 1], sc["path_rule"][p:p + 1],
                         sc["path_paint"][p:p + 1], viewport=list(sc["viewport"]))
        one.plan()
        assert [int(v) for v in one.bboxes()[0]] == [int(v) for v in bb[p]]
        ref = ctx.alloc(rows * cols * 32)
        one.render(ref, _abi.OUT_FILL_F64)
        assert np.abs(got - ref.download((rows, cols, 4), np.float64)).max() <= 1e-12
        one.destroy()
    batch.destroy()


def test_convert_and_scale_into_another_buffer(S):
    from svgrasterize_amd import _abi

    ctx = S.Context.get()
    rng = np.random.default_rng(5)
    img = rng.random((37, 53, 4))
    img[..., :3] *= img[..., 3:]
    src = ctx.from_host(img)
    n_px = 37 * 53
    for ops in (1, 2, 4, 8, 1 | 2, 4 | 8, 1 | 4 | 8):
        inplace = ctx.from_host(img)
        _abi._check(ctx.lib.svgr_layer_convert(ctx.handle, inplace.handle, n_px, ops))
        dst = ctx.alloc(n_px * 32)
        _abi._check(ctx.lib.svgr_layer_convert_to(ctx.handle, dst.handle, src.handle, n_px, ops))
        assert np.array_equal(dst.download(img.shape, np.float64), inplace.download(img.shape, np.float64)), ops
        assert np.array_equal(src.download(img.shape, np.float64), img)   # the source is left alone
    inplace = ctx.from_host(img)
    _abi._check(ctx.lib.svgr_layer_scale(ctx.handle, inplace.handle, n_px * 4, 0.375))
    dst = ctx.alloc(n_px * 32)
    _abi._check(ctx.lib.svgr_layer_scale_to(ctx.handle, dst.handle, src.handle, n_px * 4, 0.375))
    assert np.array_equal(dst.download(img.shape, np.float64), inplace.download(img.shape, np.float64))
    assert np.array_equal(dst.download(img.shape, np.float64), img * 0.375)
