"""Host-side conveniences of the reference API (Path.to_svg / __repr__ / transform / is_empty, ConvexHull.path,
Scene.__repr__ / to_path, Transform.apply, Layer.background) against what the reference returned for the same inputs
(tests/golden/hostutil_kat.npz, made by oracle/gen_golden.py --only hostutil)."""
import json
import os
import warnings

import numpy as np
import pytest

from tests import svg_cases

GOLD = os.path.join(os.path.dirname(__file__), "golden")


@pytest.fixture(scope="module")
def kat():
    z = np.load(os.path.join(GOLD, "hostutil_kat.npz"))
    return z, json.loads(str(z["meta"]))


def _packed(path):
    from svgrasterize_amd.geometry import PATH_ARC

    types, params, sizes = [], [], []
    for sub in path.subpaths:
        sizes.append(len(sub))
        for kind, args in sub:
            assert kind != PATH_ARC
            flat = np.asarray(args, dtype=np.float64).ravel()
            types.append(kind)
            params.append(np.concatenate([flat, np.zeros(8 - flat.size)]))
    return np.array(types), np.array(params).reshape(-1, 8), np.array(sizes)


def test_path_text_forms_and_host_transform(kat):
    from svgrasterize_amd import Path, Transform

    z, meta = kat
    tr = Transform(np.vstack([np.array(meta["tr"]).reshape(2, 3), [0, 0, 1]]))
    for idx, want in enumerate(meta["paths"]):
        path = Path.from_svg(want["d"])
        assert path.is_empty() == want["empty"]
        assert path.to_svg() == want["to_svg"], want["d"]
        assert repr(path) == want["repr"], want["d"]
        moved = path.transform(tr)
        types, params, sizes = _packed(moved)
        assert np.array_equal(types, z[f"p{idx}_types"]) and np.array_equal(sizes, z[f"p{idx}_sizes"])
        assert np.allclose(params, z[f"p{idx}_params"], rtol=0, atol=1e-12)
        assert moved.to_svg() == want["moved_to_svg"]
    assert Path([]).is_empty() and repr(Path([])) == "EMPTY" and Path([]).to_svg() == ""
    pts = np.array([[1.0, 2.0], [-3.5, 0.25]])
    assert np.array_equal(tr.apply()(pts), tr(pts))


def test_hull_outline(kat):
    from svgrasterize_amd import ConvexHull

    _z, meta = kat
    hull = ConvexHull([[0, 0], [4, 0], [4, 3], [2, 1], [0, 3], [2, 5]])
    assert [[float(x) for x in p] for p in hull.points] == meta["hull"]["points"]
    assert repr(hull.path()) == meta["hull"]["repr"]


def test_scene_repr_and_to_path(kat):
    from svgrasterize_amd import Transform, svg_scene_from_str
    from svgrasterize_amd.scenedump import gather_path

    z, meta = kat
    cases = {name: (text, width) for name, text, width in svg_cases.CASES}
    view = Transform().matrix(0, 1, 0, 1, 0, 0).scale(0.5)
    for idx, want in enumerate(meta["scenes"]):
        text, width = cases[want["name"]]
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            scene, _ids, _size = svg_scene_from_str(text, width=width)
        assert repr(scene) == want["repr"], want["name"]
        lines, cubics = gather_path(scene.to_path(view))
        assert lines.shape == z[f"s{idx}_lines"].shape and cubics.shape == z[f"s{idx}_cubics"].shape, want["name"]
        assert np.allclose(lines, z[f"s{idx}_lines"], rtol=0, atol=1e-9), want["name"]
        assert np.allclose(cubics, z[f"s{idx}_cubics"], rtol=0, atol=1e-9), want["name"]


@pytest.mark.gpu
def test_layer_background(kat):
    import svgrasterize_amd as S

    S.Context.get()
    z, meta = kat
    colour = np.array(meta["bg"]["colour"])
    for idx, (pre, lin) in enumerate(meta["bg"]["flags"]):
        out = S.Layer(z[f"bg{idx}_in"], (2, 5), pre_alpha=pre, linear_rgb=lin).background(colour)
        assert out.pre_alpha and out.linear_rgb and tuple(out.offset) == (2, 5)
        assert np.allclose(out.image, z[f"bg{idx}_out"], rtol=0, atol=1e-15)


def test_default_strip_bands_one_strip_per_rank():
    """The bench's sharding default: one strip per rank, never under 128 scanlines; every band has one owner."""
    from svgrasterize_amd import dist as sdist

    assert sdist.default_strip_bands(8192, 16, 8) == 64 and sdist.default_strip_bands(8192, 16, 2) == 256
    assert sdist.default_strip_bands(4096, 16, 8) == 32 and sdist.default_strip_bands(512, 16, 8) == 4   # (8 would idle four ranks)
    assert sdist.default_strip_bands(4096, 16, 3) == 86
    assert sdist.default_strip_bands(17 * 16, 16, 8) == 2      # (rounding up: strips of 3 = 6 strips for 8 ranks)
    for rows, world in ((8192, 8), (4096, 3), (1000, 4), (100, 2), (17 * 16, 8), (512, 8), (40, 8), (16 * 9, 8)):
        strip = sdist.default_strip_bands(rows, 16, world)
        nb = sdist.n_bands(rows, 16)
        owned = [sdist.owned_bands(rows, 16, r, world, strip) for r in range(world)]
        owners = [sum(b in o for o in owned) for b in range(nb)]
        assert owners == [1] * nb                                   # every band has one owner
        assert sum(1 for o in owned if o) == min(world, nb), (rows, world, strip)   # and no rank idles while there are bands to own


def test_batch_routing_of_scene_nodes_is_host_logic():
    """Which Scene nodes become entries of ONE device batch (scene._batchable_leaves) and how a run is tidied before it is
    packed (effective_bboxes, _drop_empty): plain host logic, checked without a GPU."""
    import svgrasterize_amd as S
    from svgrasterize_amd import scene as sc

    sq = lambda x, y, w: S.Path.from_svg(f"M{x},{y} h{w} v{w} h{-w} Z")
    red, blue = np.array([0.5, 0.0, 0.0, 0.5]), np.array([0.0, 0.0, 1.0, 1.0])
    lin = S.GradLinear(np.array([0.0, 0.0]), np.array([10.0, 0.0]), [(0.0, red), (1.0, blue)], None, "pad", False, None)
    tr = S.Transform().matrix(0, 1, 0, 1, 0, 0)
    fill = S.Scene.fill
    # plain leaves, an opacity over a leaf, a gradient leaf: one entry each; the opacity lands in the paint / the multiplier
    leaves = sc._batchable_leaves(S.Scene.group([fill(sq(0, 0, 4), red), fill(sq(2, 2, 4), blue).opacity(0.5), fill(sq(1, 1, 3), lin).opacity(0.25)]), tr, True)
    assert [l[4] for l in leaves] == [0, 0, 0] and [l[5] for l in leaves] == [None] * 3
    assert np.allclose(leaves[1][3], blue * 0.5) and leaves[0][6] is None and leaves[1][6] is None
    assert leaves[2][6] is not None and np.allclose(leaves[2][3], 0.25) and leaves[2][6][0].kind == 1
    # CLIP of a single leaf: clip source + clipped entry; CLIP / OPACITY over a group: members tagged with one group
    one = sc._batchable_leaves(fill(sq(0, 0, 8), red).clip(fill(sq(2, 2, 3), blue)), tr, True)
    assert [l[4] for l in one] == [1, 2] and one[1][5] is None
    grp = S.Scene.group([fill(sq(0, 0, 8), red), fill(sq(1, 1, 2), lin)])
    clipped = sc._batchable_leaves(grp.clip(fill(sq(2, 2, 3), blue)), tr, True)
    assert [l[4] for l in clipped] == [1, 0, 0] and clipped[1][5] is clipped[2][5] and clipped[1][5][1:] == (1.0, True)
    faded = sc._batchable_leaves(grp.opacity(0.3), tr, True)
    assert [l[4] for l in faded] == [0, 0] and faded[0][5] is faded[1][5] and faded[0][5][1:] == (0.3, False)
    # what stays on the per-node route
    bbox_grad = S.GradLinear(np.array([0.0, 0.0]), np.array([1.0, 0.0]), [(0.0, red), (1.0, blue)], None, "pad", True, None)
    # (an objectBoundingBox gradient is a batch entry whose frame is still to come -- under a transform that keeps the axes apart)
    pending = sc._batchable_leaves(fill(sq(0, 0, 4), bbox_grad), tr, True)
    assert len(pending) == 1 and pending[0][6][0] is None and pending[0][6][2] is bbox_grad
    assert sc._batchable_leaves(fill(sq(0, 0, 4), bbox_grad), tr.rotate(0.3), True) is None
    assert sc._axes_apart(tr) and sc._axes_apart(tr.scale(3.0, -0.5).translate(7, 9)) and not sc._axes_apart(tr.skew(0.1, 0.0))
    for node in (S.Scene.group([grp.opacity(0.5), fill(sq(0, 0, 2), red)]).opacity(0.5),  # a group inside a group's opacity
                 grp.clip(S.Scene.group([fill(sq(0, 0, 2), red), fill(sq(1, 1, 2), red)])),  # a clip that is not one path
                 fill(sq(0, 0, 4), red).clip(fill(sq(0, 0, 2), red), bbox_units=True)):
        assert sc._batchable_leaves(node, tr, True) is None
    with pytest.raises(ValueError):
        sc._batchable_leaves(fill(sq(0, 0, 4), red, "winding"), tr, True)
    # effective bboxes: a clipped fill / a member of a clipped group is cut to the clip's bbox; an empty clip hides them
    run = clipped + one
    boxes = [(2, 2, 3, 3), (0, 0, 8, 8), (1, 1, 2, 2), (10, 10, 0, 0), (0, 0, 8, 8)]
    assert sc.effective_bboxes(run, boxes) == [(2, 2, 3, 3), (2, 2, 1, 1), None]
    # leaves without segments go, and a clip source goes with everything it clips (and the other way round)
    empty = S.Path([])
    tag = clipped[1][5]
    run = [sc._leaf(empty, tr.m6(), 0, np.zeros(4), 1), sc._leaf(sq(0, 0, 2), tr.m6(), 0, red, 0, tag), sc._leaf(sq(0, 0, 3), tr.m6(), 0, red),
           sc._leaf(sq(0, 0, 2), tr.m6(), 0, np.zeros(4), 1), sc._leaf(empty, tr.m6(), 0, red, 2), sc._leaf(empty, tr.m6(), 0, red)]
    kept = sc._drop_empty(run)
    assert len(kept) == 1 and kept[0] is run[2]


def test_two_threads_share_nothing_but_the_lock(monkeypatch):
    """The walk state is per thread, the serial numbers come from one locked counter, and top-level renders of a process run
    one after the other (`_state.RENDER_LOCK`: the library has no lock and all threads share one context and stream)."""
    import threading
    import time

    from svgrasterize_amd import Path, Scene, Transform, _state, scene as scene_mod

    serials, seen_state, inside, overlap = [], [], [0], [0]
    guard = threading.Lock()

    def fake_render(self, transform, mask_only=False, viewport=None, linear_rgb=False, _asked=False):
        with guard:
            inside[0] += 1
            overlap[0] = max(overlap[0], inside[0])
        seen_state.append((threading.get_ident(), _state.STATE.serial, _state.STATE.leaf_memo is not None))
        for _ in range(200):
            serials.append(scene_mod._new_group(1.0, False)[0])
        time.sleep(0.02)
        with guard:
            inside[0] -= 1
        return None

    from svgrasterize_amd import displaylist

    monkeypatch.setattr(displaylist, "ENABLED", False)   # (the walk is what is under test: a scene of fills alone would be drawn from its display list)
    monkeypatch.setattr(Scene, "_render", fake_render)
    monkeypatch.setattr(scene_mod, "_collect_mask_jobs", lambda *a, **k: None)
    sc = Scene.fill(Path.from_svg("M1,1 L5,1 L3,4 Z"), np.array([1.0, 0.0, 0.0, 1.0])).opacity(0.5)
    tr = Transform()
    errors = []

    def work():
        try:
            for _ in range(3):
                sc.render(tr, viewport=[0, 0, 8, 8])
        except Exception as exc:  # noqa: BLE001
            errors.append(exc)

    threads = [threading.Thread(target=work) for _ in range(4)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors
    assert overlap[0] == 1, "two top-level renders ran at the same time"
    assert len(serials) == len(set(serials)) == 4 * 3 * 200, "a group serial was handed out twice"
    assert len({s for _, s, _ in seen_state}) == 12 and all(memo for _, _, memo in seen_state)
    assert _state.STATE.leaf_memo is None and _state.STATE.retain is None   # (nothing left behind in this thread)
