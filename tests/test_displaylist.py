"""A scene that is batch entries only, compiled once to flat arrays (svgrasterize.py_amd/displaylist.py): the arrays a render draws
from must be, leaf for leaf and bit for bit, what `Scene.render`'s walk (S:649-752 as `scene._batchable_leaves_` + `_drop_empty` +
`build_batch` restate it) hands to the device -- the chain products of stacked 3x3 matmuls included."""
import os

import numpy as np
import pytest

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")


def _walk_arrays(sc_mod, scene, tr, lin):
    leaves = sc_mod._batchable_leaves(scene, tr, lin)
    if leaves is None:
        return None
    leaves = sc_mod._drop_empty(leaves)
    m6 = np.array([l[1] for l in leaves]).reshape(len(leaves), 6)
    paints = np.array([l[3] for l in leaves]).reshape(len(leaves), 4)
    rules = np.array([l[2] | (l[4] << 1) for l in leaves], dtype=np.uint8)
    groups = [None if l[5] is None else (l[5][0], l[5][1], l[5][2]) for l in leaves]
    return leaves, m6, paints, rules, groups


@pytest.mark.parametrize("name", ["tiger", "material"])
@pytest.mark.parametrize("lin", [False, True])
def test_display_list_is_the_walk(name, lin):
    import svgrasterize_amd as S
    from svgrasterize_amd import displaylist, scene as sc_mod, scenedump

    scene, _info, _z = scenedump.load_scene(os.path.join(GOLDEN, f"scene_{name}.npz"))
    dl = displaylist._compile(scene, lin)
    assert dl is not None
    for tr in (S.Transform().matrix(0, 1, 0, 1, 0, 0), S.Transform().matrix(0, 1, 0, 1, 0, 0).scale(0.37).rotate(0.3).translate(11.5, -3.25)):
        leaves, m6, paints, rules, groups = _walk_arrays(sc_mod, scene, tr, lin)
        assert dl.n == len(leaves) and all(a[0] is b for a, b in zip(leaves, dl.paths))
        assert np.array_equal(dl.matrices(tr), m6), "chain products differ from the walk's Transform.__matmul__"
        assert np.array_equal(dl.current_paints(), paints)
        assert np.array_equal(dl.rules, rules)
        # isolated groups: same members, same opacity / clipped, same clip source position
        if dl.group_src is None:
            assert all(g is None for g in groups)
        else:
            seen = {}
            for i, g in enumerate(groups):
                if g is None:
                    assert dl.path_group[i] == -1
                    continue
                gid = seen.setdefault(g[0], len(seen))
                assert dl.path_group[i] == gid and dl.group_op[gid] == g[1]
                assert (dl.group_src[gid] >= 0) == g[2]


def test_display_list_refuses_what_is_not_flat_and_rereads_paints():
    import svgrasterize_amd as S
    from svgrasterize_amd import displaylist, scenedump

    scene, _info, _z = scenedump.load_scene(os.path.join(GOLDEN, "scene_icons4096.npz"))
    assert displaylist._compile(scene, False) is None          # gradients, filters: the general route
    p = S.Path.from_svg("M1,1 H9 V9 H1 Z")
    col = np.array([0.5, 0.25, 0.0, 0.5])
    sc = S.Scene.group([S.Scene.fill(p, col), S.Scene.fill(p, np.array([0.0, 0.0, 0.2, 0.2])).transform(S.Transform().translate(3, 4))])
    dl = displaylist.get(sc, False)
    assert dl is displaylist.get(sc, False) and dl.n == 2
    a = dl.current_paints().copy()
    col[0] = 0.125                                             # edited in place: drawn with the new value, as the reference would
    b = dl.current_paints()
    assert not np.array_equal(a, b) and np.array_equal(b[0], S.geometry.solid_paint(col, False))
    with pytest.raises(ValueError):
        displaylist.get(S.Scene.group([S.Scene.fill(p, col, "winding"), S.Scene.fill(p, col)]), False)   # S:989
    empty = S.Scene.group([S.Scene.fill(S.Path([]), col), S.Scene.fill(S.Path([]), col)])
    assert displaylist.get(empty, False).n == 0


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["tiger", "material"])
def test_display_list_render_is_the_walks_render(name, monkeypatch):
    """The layer drawn from the display list against the layer the general route draws (SVGR_NO_DISPLAY_LISTS): same offset and
    shape, same pixels (1e-12: the order of the LDS atomics), same hull."""
    import svgrasterize_amd as S
    from svgrasterize_amd import displaylist, scenedump

    scene, info, _z = scenedump.load_scene(os.path.join(GOLDEN, f"scene_{name}.npz"))
    size = 1024
    tr = S.Transform().matrix(0, 1, 0, 1, 0, 0).scale(size / info["full"]["size"][0])
    vp = [0, 0, size, size]
    monkeypatch.setattr(displaylist, "ENABLED", True)
    a_layer, a_hull = scene.render(tr, viewport=vp, linear_rgb=False)
    monkeypatch.setattr(displaylist, "ENABLED", False)
    b_layer, b_hull = scene.render(tr, viewport=vp, linear_rgb=False)
    assert tuple(int(v) for v in a_layer.offset) == tuple(int(v) for v in b_layer.offset)
    assert a_layer.image.shape == b_layer.image.shape and (a_layer.pre_alpha, a_layer.linear_rgb) == (b_layer.pre_alpha, b_layer.linear_rgb)
    assert np.abs(a_layer.image - b_layer.image).max() <= 1e-12 and a_layer.image.any()
    assert np.array_equal(np.asarray(a_hull.points), np.asarray(b_hull.points))
    # a window of it, and nothing at all
    sub = [200, 300, 256, 128]
    monkeypatch.setattr(displaylist, "ENABLED", True)
    a2 = scene.render(tr, viewport=sub, linear_rgb=True)
    monkeypatch.setattr(displaylist, "ENABLED", False)
    b2 = scene.render(tr, viewport=sub, linear_rgb=True)
    assert (a2 is None) == (b2 is None)
    if a2 is not None:
        assert tuple(int(v) for v in a2[0].offset) == tuple(int(v) for v in b2[0].offset)
        assert np.abs(a2[0].image - b2[0].image).max() <= 1e-12
    monkeypatch.setattr(displaylist, "ENABLED", True)
    assert scene.render(tr, viewport=[5000, 5000, 64, 64], linear_rgb=False) is None
