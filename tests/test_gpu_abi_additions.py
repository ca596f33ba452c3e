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


# ---- ABI 5: svgr_layer_compose_over, svgr_layer_convert_scale_to, svgr_layer_convolve_ops, against the calls they stand in for ----
def _rand_layer(rng, rows, cols, ch=4):
    img = rng.random((rows, cols, ch))
    if ch == 4:
        img[..., :3] *= img[..., 3:]
    return img


def _bb(r0, c0, rows, cols):
    import ctypes as C

    return (C.c_int64 * 4)(r0, c0, rows, cols)


def test_compose_over_in_one_pass_is_the_pass_per_layer(S):
    """svgr_layer_compose_over (converting sources as it reads them) = convert each source, clear the union, svgr_layer_over one
    layer after the other: the same bits, for 3 and for 30 layers (more than one launch's table), 1- and 4-channel sources."""
    import ctypes as C

    from svgrasterize_amd import _abi

    ctx = S.Context.get()
    rng = np.random.default_rng(11)
    for n in (1, 3, 30):
        specs = []
        for i in range(n):
            rows, cols = int(rng.integers(5, 60)), int(rng.integers(5, 70))
            r0, c0 = int(rng.integers(-20, 40)), int(rng.integers(-30, 50))
            ch = 1 if i % 5 == 3 else 4
            ops = 0 if ch == 1 else int(rng.choice([0, 0, 8, 4 | 8, 2 | 8, 1 | 2 | 8]))
            img = _rand_layer(rng, rows, cols, ch)
            if ops & 8 and not ops & 1:
                img = rng.random((rows, cols, 4))   # (a straight-alpha source)
            specs.append((img, (r0, c0, rows, cols), ch, ops))
        u0 = min(s[1][0] for s in specs); v0 = min(s[1][1] for s in specs)
        u1 = max(s[1][0] + s[1][2] for s in specs); v1 = max(s[1][1] + s[1][3] for s in specs)
        shape = (u1 - u0, v1 - v0, 4)
        obb = _bb(u0, v0, shape[0], shape[1])
        # the pass per layer
        ref = ctx.alloc(shape[0] * shape[1] * 32)
        ref.zero()
        for i, (img, bb, ch, ops) in enumerate(specs):
            b = ctx.from_host(img)
            if ops:
                _abi._check(ctx.lib.svgr_layer_convert(ctx.handle, b.handle, bb[2] * bb[3], ops))
            _abi._check(ctx.lib.svgr_layer_over(ctx.handle, ref.handle, obb, b.handle, _bb(*bb), ch, int(i == 0)))
        # one pass
        out = ctx.alloc(shape[0] * shape[1] * 32)
        srcs = [ctx.from_host(s[0]) for s in specs]
        handles = (_abi._P * n)(*[b.handle for b in srcs])
        bbs = (C.c_int64 * (4 * n))(*[v for s in specs for v in s[1]])
        chs = (C.c_int32 * n)(*[s[2] for s in specs])
        opsa = (C.c_uint32 * n)(*[s[3] for s in specs])
        # (the output need not be cleared: poison it)
        out.upload(np.full(shape, np.nan))
        _abi._check(ctx.lib.svgr_layer_compose_over(ctx.handle, out.handle, obb, n, handles, bbs, chs, opsa))
        assert np.array_equal(out.download(shape, np.float64), ref.download(shape, np.float64)), n
    with pytest.raises(ValueError):   # ops on a 1-channel source
        one = ctx.from_host(np.zeros((4, 4, 1)))
        _abi._check(ctx.lib.svgr_layer_compose_over(ctx.handle, out.handle, obb, 1, (_abi._P * 1)(one.handle), (C.c_int64 * 4)(0, 0, 4, 4),
                                                    (C.c_int32 * 1)(1), (C.c_uint32 * 1)(8)))


def test_convert_and_scale_in_one_pass(S):
    from svgrasterize_amd import _abi

    ctx = S.Context.get()
    rng = np.random.default_rng(12)
    img = _rand_layer(rng, 41, 29)
    n_px = 41 * 29
    src = ctx.from_host(img)
    for ops in (0, 1, 8, 1 | 2 | 8, 1 | 4 | 8):
        two = ctx.from_host(img)
        _abi._check(ctx.lib.svgr_layer_convert(ctx.handle, two.handle, n_px, ops))
        _abi._check(ctx.lib.svgr_layer_scale(ctx.handle, two.handle, n_px * 4, 0.3125))
        one = ctx.alloc(n_px * 32)
        _abi._check(ctx.lib.svgr_layer_convert_scale_to(ctx.handle, one.handle, src.handle, n_px, ops, 0.3125))
        assert np.array_equal(one.download(img.shape, np.float64), two.download(img.shape, np.float64)), ops


def test_blur_of_a_source_that_still_needs_its_conversion(S):
    """svgr_layer_convolve_ops = svgr_layer_convert, then svgr_layer_convolve: the same bits (separable kernels in the two blocked
    passes, a small non-separable one in the kernel argument, a large one through the uploaded stencil)."""
    from svgrasterize_amd import _abi

    ctx = S.Context.get()
    rng = np.random.default_rng(13)
    img = _rand_layer(rng, 57, 83)
    rows, cols = 57, 83

    def gauss(n, s):
        x = np.arange(n) - (n - 1) / 2
        g = np.exp(-x * x / (2 * s * s))
        return g / g.sum()

    kernels = [np.outer(gauss(9, 1.7), gauss(13, 2.4)), np.outer(gauss(45, 7.0), gauss(45, 7.0)), rng.random((3, 5)), rng.random((15, 17)),
               np.outer(gauss(1, 1.0), gauss(7, 1.1)), np.outer(gauss(5, 0.9), gauss(1, 1.0))]
    for k in kernels:
        k = np.ascontiguousarray(k, dtype=np.float64)
        kw, kh = k.shape
        oshape = (rows + kw - 1, cols + kh - 1, 4)
        for ops in (0, 1 | 2):
            conv = ctx.from_host(img)
            if ops:
                _abi._check(ctx.lib.svgr_layer_convert(ctx.handle, conv.handle, rows * cols, ops))
            ref = ctx.alloc(oshape[0] * oshape[1] * 32)
            _abi._check(ctx.lib.svgr_layer_convolve(ctx.handle, ref.handle, conv.handle, rows, cols, _abi.ptr(k), kw, kh))
            src = ctx.from_host(img)
            out = ctx.alloc(oshape[0] * oshape[1] * 32)
            _abi._check(ctx.lib.svgr_layer_convolve_ops(ctx.handle, out.handle, src.handle, rows, cols, _abi.ptr(k), kw, kh, ops))
            assert np.array_equal(out.download(oshape, np.float64), ref.download(oshape, np.float64)), (k.shape, ops)
            assert np.array_equal(src.download(img.shape, np.float64), img)
            # ... and both are the full 2-D convolution (S:106-118) of the converted image
            if kw * kh <= 255:
                from scipy.signal import convolve as sp_convolve

                want = sp_convolve(conv.download(img.shape, np.float64), k[..., None], mode="full", method="direct")
                assert np.abs(out.download(oshape, np.float64) - want).max() <= 1e-14 * max(np.abs(want).max(), 1.0) * (kw * kh) ** 0.5


def test_a_noted_conversion_is_the_conversion(S):
    """Layer.convert of a device layer only notes the ops (`_ops`): whoever reads the layer -- `.image`, compose, opacity, another
    convert that cannot be folded -- sees the converted pixels, bit for bit what the eager conversion gave."""
    from svgrasterize_amd import _abi

    ctx = S.Context.get()
    rng = np.random.default_rng(14)
    img = _rand_layer(rng, 33, 47)

    def dev_layer():
        return S.Layer._from_device(ctx.from_host(img), img.shape, (3, 5), True, False)

    def eager(ops):
        b = ctx.from_host(img)
        _abi._check(ctx.lib.svgr_layer_convert(ctx.handle, b.handle, 33 * 47, ops))
        return b.download(img.shape, np.float64)

    lazy = dev_layer().convert(pre_alpha=False, linear_rgb=True)
    assert lazy._ops == 3 and lazy.pre_alpha is False and lazy.linear_rgb is True
    assert np.array_equal(lazy.image, eager(1 | 2))
    # folded: straight linear -> premultiplied linear on top of the note (1, 2, then 8)
    twice = dev_layer().convert(pre_alpha=False, linear_rgb=True).convert(pre_alpha=True, linear_rgb=True)
    assert twice._ops == (1 | 2 | 8)
    assert np.array_equal(twice.image, eager(1 | 2 | 8))
    # not foldable: back to sRGB needs "premultiplied -> straight" behind "straight -> premultiplied"
    back = dev_layer().convert(pre_alpha=False, linear_rgb=True).convert(pre_alpha=True, linear_rgb=True).convert(pre_alpha=True, linear_rgb=False)
    first = ctx.from_host(eager(1 | 2 | 8))
    _abi._check(ctx.lib.svgr_layer_convert(ctx.handle, first.handle, 33 * 47, 1 | 4 | 8))
    assert np.array_equal(back.image, first.download(img.shape, np.float64))
    # opacity and compose read through the note
    faded = dev_layer().convert(pre_alpha=False, linear_rgb=True).opacity(0.25, linear_rgb=True)
    assert np.array_equal(faded.image, eager(1 | 2 | 8) * 0.25)
    other = S.Layer(_rand_layer(rng, 20, 60), (0, 0), True, True)
    got = S.Layer.compose([other, dev_layer().convert(pre_alpha=False, linear_rgb=True)], linear_rgb=True)
    want = S.Layer.compose([other, S.Layer(eager(1 | 2), (3, 5), False, True)], linear_rgb=True)
    assert got.offset == want.offset and np.array_equal(got.image, want.image)
    # the source layer is untouched by its converted views
    src = dev_layer()
    view = src.convert(pre_alpha=False, linear_rgb=True)
    _ = view.image
    assert np.array_equal(src.image, img)


@pytest.mark.parametrize("what", ["clips", "groups", "gradients"])
def test_windows_in_one_launch_for_every_variant_and_both_canvas_types(S, what):
    """The window table serves every variant of the tile kernel but the production one: batches with clip pairs, with isolated
    groups, with gradient entries, float32 and float64 canvases -- each against the same windows drawn a launch each."""
    from svgrasterize_amd import _abi
    from svgrasterize_amd import scene as sm

    ctx = S.Context.get()
    rng = np.random.default_rng(21)

    def blob(cx, cy, r):
        k = 0.5522847498 * r
        return S.Path([[(S.PATH_CUBIC, [[cx + r, cy], [cx + r, cy + k], [cx + k, cy + r], [cx, cy + r]]),
                        (S.PATH_CUBIC, [[cx, cy + r], [cx - k, cy + r], [cx - r, cy + k], [cx - r, cy]]),
                        (S.PATH_CUBIC, [[cx - r, cy], [cx - r, cy - k], [cx - k, cy - r], [cx, cy - r]]),
                        (S.PATH_CUBIC, [[cx, cy - r], [cx + k, cy - r], [cx + r, cy - k], [cx + r, cy]])]])

    def colour():
        c = rng.random(4)
        c[:3] *= c[3]
        return c

    nodes = []
    for i in range(40):
        cx, cy, r = rng.uniform(30, 370), rng.uniform(30, 370), rng.uniform(15, 60)
        node = S.Scene.fill(blob(cx, cy, r), colour())
        if what == "clips" and i % 3 == 0:
            node = node.clip(S.Scene.fill(blob(cx + 10, cy - 5, r * 0.7), colour()))
        if what == "groups" and i % 4 == 0:
            node = S.Scene.group([node, S.Scene.fill(blob(cx + 12, cy + 9, r * 0.8), colour())]).opacity(0.6)
        if what == "gradients" and i % 3 == 0:
            grad = S.GradLinear(np.array([cx - r, cy]), np.array([cx + r, cy + r]), [(0.0, colour()), (1.0, colour())], None, "reflect", False, None)
            node = S.Scene.fill(blob(cx, cy, r), grad)
        nodes.append(node)
    tr = S.Transform().matrix(0, 1, 0, 1, 0, 0)
    leaves = sm._batchable_leaves(S.Scene.group(nodes), tr, True)
    assert leaves is not None
    batch = sm.build_batch(leaves, [0, 0, 400, 400], ctx)
    batch.plan()
    wins = [(0, 0, 400, 400), (13, 27, 200, 90), (100, 100, 64, 64), (350, 10, 50, 380), (1, 399, 398, 1), (200, 0, 16, 400)]
    for kind, dt, px, tol in ((_abi.OUT_CANVAS_F64, np.float64, 32, 1e-12), (_abi.OUT_CANVAS_F32, np.float32, 16, 2.0 ** -22)):
        one = []
        for w in wins:
            o = ctx.alloc(w[2] * w[3] * px)
            batch.render(o, kind, 0, window=w)
            one.append(o.download((w[2], w[3], 4), dt))
        outs = [ctx.alloc(w[2] * w[3] * px) for w in wins]
        batch.render_windows(outs, kind, wins)
        for w, o, ref in zip(wins, outs, one):
            got = o.download((w[2], w[3], 4), dt)
            assert np.abs(got.astype(np.float64) - ref.astype(np.float64)).max() <= tol, (what, kind, w)
        assert any(r.any() for r in one)
    batch.destroy()


def test_compose_in_in_one_pass_is_crop_then_in(S):
    """svgr_layer_compose_in = convert each source, svgr_layer_crop4 of the first, svgr_layer_in of the others: the same bits
    (a 1-channel mask first -- the clip route --, 4-channel layers, sources that still need a conversion, 30 layers)."""
    import ctypes as C

    from svgrasterize_amd import _abi

    ctx = S.Context.get()
    rng = np.random.default_rng(31)
    for n, first_ch in ((2, 1), (2, 4), (3, 1), (30, 4)):
        specs = []
        for i in range(n):
            rows, cols = int(rng.integers(40, 80)), int(rng.integers(40, 90))
            r0, c0 = int(rng.integers(-10, 10)), int(rng.integers(-10, 10))
            ch = first_ch if i == 0 else (1 if i % 7 == 5 else 4)
            ops = 0 if ch == 1 else int(rng.choice([0, 8, 1 | 2 | 8, 4 | 8]))
            specs.append((_rand_layer(rng, rows, cols, ch), (r0, c0, rows, cols), ch, ops))
        u0 = max(s[1][0] for s in specs); v0 = max(s[1][1] for s in specs)
        u1 = min(s[1][0] + s[1][2] for s in specs); v1 = min(s[1][1] + s[1][3] for s in specs)
        assert u1 > u0 and v1 > v0
        shape = (u1 - u0, v1 - v0, 4)
        obb = _bb(u0, v0, shape[0], shape[1])
        ref = ctx.alloc(shape[0] * shape[1] * 32)
        for i, (img, bb, ch, ops) in enumerate(specs):
            b = ctx.from_host(img)
            if ops:
                _abi._check(ctx.lib.svgr_layer_convert(ctx.handle, b.handle, bb[2] * bb[3], ops))
            if i == 0:
                _abi._check(ctx.lib.svgr_layer_crop4(ctx.handle, ref.handle, obb, b.handle, _bb(*bb), ch))
            else:
                _abi._check(ctx.lib.svgr_layer_in(ctx.handle, ref.handle, obb, b.handle, _bb(*bb), ch))
        out = ctx.alloc(shape[0] * shape[1] * 32)
        out.upload(np.full(shape, np.nan))
        srcs = [ctx.from_host(s[0]) for s in specs]
        _abi._check(ctx.lib.svgr_layer_compose_in(ctx.handle, out.handle, obb, n, (_abi._P * n)(*[b.handle for b in srcs]),
                                                  (C.c_int64 * (4 * n))(*[v for s in specs for v in s[1]]), (C.c_int32 * n)(*[s[2] for s in specs]),
                                                  (C.c_uint32 * n)(*[s[3] for s in specs])))
        assert np.array_equal(out.download(shape, np.float64), ref.download(shape, np.float64)), (n, first_ch)
    # through Layer.compose: a mask and a device layer with a noted conversion
    mask = S.Layer(rng.random((50, 60, 1)), (2, 3), True, True)
    img = _rand_layer(rng, 55, 58)
    dev = S.Layer._from_device(ctx.from_host(img), img.shape, (0, 0), True, False).convert(pre_alpha=False, linear_rgb=True)
    eager = ctx.from_host(img)
    _abi._check(ctx.lib.svgr_layer_convert(ctx.handle, eager.handle, 55 * 58, 1 | 2))
    got = S.Layer.compose([mask, dev], S.COMPOSE_IN, linear_rgb=True)
    want = S.Layer.compose([mask, S.Layer(eager.download(img.shape, np.float64), (0, 0), False, True)], S.COMPOSE_IN, linear_rgb=True)
    assert got.offset == want.offset and np.array_equal(got.image, want.image)


def test_extents_are_the_bounds_of_all_flattened_points(S):
    """svgr_batch_get_extents: per path the exact min / max of every flattened point, whatever the viewport clips -- the numbers
    ConvexHull(lines).bbox is made from (S:993, S:2010-2016) -- against the unculled edge list (svgr_batch_all_edges); only between
    the plan and the first render."""
    from svgrasterize_amd import _abi, synth

    sc = synth.make_scene(512, 96)
    ctx = S.Context.get()
    for vp in ([0, 0, 512, 512], [100, 60, 200, 300]):   # (the second viewport cuts most paths)
        batch = _abi.Batch(ctx, sc["segs"], sc["seg_kind"], sc["path_seg_off"], sc["path_m6"], sc["path_rule"], sc["path_paint"], viewport=vp)
        batch.plan()
        ext = batch.extents()
        edges, edge_path = batch.all_edges()
        pts = edges.reshape(-1, 2, 2)
        for p in range(batch.n_paths):
            mine = pts[edge_path == p].reshape(-1, 2)
            if len(mine) == 0:
                assert np.isinf(ext[p]).all()
                continue
            want = np.array([mine[:, 0].min(), mine[:, 1].min(), mine[:, 0].max(), mine[:, 1].max()])
            assert np.array_equal(ext[p], want), p
        batch.plan()   # (all_edges left the counter arena dirty: plan again for a render)
        out = ctx.alloc(vp[2] * vp[3] * 32)
        batch.render(out, _abi.OUT_CANVAS_F64)
        with pytest.raises(_abi.SvgrError):
            batch.extents()   # (the keys are the plan's pass's: gone with the first render)
        batch.destroy()
