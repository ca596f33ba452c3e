"""Frozen synthetic many-path scenes: the bench workload and the large parity cases.

The reference ships no generator; this one is specified completely so that every run, rank and
round renders the same scene:

* PRNG splitmix64 seeded 0x5F3759DF; ``U() = (next() >> 11) * 2**-53``.
* canvas S x S; for path i = 0..N-1 the draws are consumed in exactly this order:
  centre ``(U*S, U*S)``; ``r = 32 + 224*U**2`` pixels (absolute, not scaled with S);
  ``k = 3 + floor(6*U)`` closed cubic segments; for j in 0..k-1: anchor j at angle
  ``2*pi*(j + 0.8*(U - 0.5))/k`` and radius ``r*(0.5 + 0.5*U)``; then for each segment j its two
  control points = start anchor + ``0.6*r*(2U-1, 2U-1)`` and end anchor + ``0.6*r*(2U-1, 2U-1)``;
  colour straight RGB ``U, U, U``, alpha ``0.25 + 0.75*U``, premultiplied, taken as already in the
  compositing space; fill rule evenodd when ``i % 8 == 7`` else nonzero; paint order = i.
* points are (x, y) in user space and go through the CLI's x/y swap transform
  ``matrix(0, 1, 0, 1, 0, 0)`` (reference S:3823), i.e. row = y, col = x.
"""
from __future__ import annotations

import math

import numpy as np

SEED = 0x5F3759DF
_GOLDEN = 0x9E3779B97F4A7C15
_M1 = 0xBF58476D1CE4E5B9
_M2 = 0x94D049BB133111EB
SWAP_M6 = np.array([0.0, 1.0, 0.0, 1.0, 0.0, 0.0])


def splitmix64_uniform(n: int, seed: int = SEED) -> np.ndarray:
    """First n outputs of splitmix64(seed) mapped to [0, 1) doubles."""
    with np.errstate(over="ignore"):
        idx = np.arange(1, n + 1, dtype=np.uint64)
        z = np.uint64(seed) + idx * np.uint64(_GOLDEN)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(_M1)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(_M2)
        z = z ^ (z >> np.uint64(31))
    return (z >> np.uint64(11)).astype(np.float64) * 2.0 ** -53


def make_scene(size: int, n_paths: int, seed: int = SEED):
    """Returns dict(segs (n,8) cubics in user space, seg_kind, path_seg_off, path_m6, path_rule,
    path_paint, viewport)."""
    u = splitmix64_uniform(n_paths * 64 + 64, seed)
    pos = 0

    def U():
        nonlocal pos
        v = u[pos]
        pos += 1
        return float(v)

    segs, offs, rules, paints = [], [0], [], []
    for i in range(n_paths):
        cx, cy = U() * size, U() * size
        uu = U()
        r = 32.0 + 224.0 * uu * uu
        k = 3 + int(math.floor(6.0 * U()))
        anchors = []
        for j in range(k):
            ang = 2.0 * math.pi * (j + 0.8 * (U() - 0.5)) / k
            rad = r * (0.5 + 0.5 * U())
            anchors.append((cx + rad * math.cos(ang), cy + rad * math.sin(ang)))
        for j in range(k):
            a, b = anchors[j], anchors[(j + 1) % k]
            c1 = (a[0] + 0.6 * r * (2.0 * U() - 1.0), a[1] + 0.6 * r * (2.0 * U() - 1.0))
            c2 = (b[0] + 0.6 * r * (2.0 * U() - 1.0), b[1] + 0.6 * r * (2.0 * U() - 1.0))
            segs.append([a[0], a[1], c1[0], c1[1], c2[0], c2[1], b[0], b[1]])
        offs.append(len(segs))
        rgb = [U(), U(), U()]
        alpha = 0.25 + 0.75 * U()
        paints.append([rgb[0] * alpha, rgb[1] * alpha, rgb[2] * alpha, alpha])
        rules.append(1 if i % 8 == 7 else 0)
    segs = np.array(segs, dtype=np.float64).reshape(-1, 8)
    return dict(
        segs=segs,
        seg_kind=np.ones(len(segs), dtype=np.uint8),
        path_seg_off=np.array(offs, dtype=np.int64),
        path_m6=np.tile(SWAP_M6, (n_paths, 1)),
        path_rule=np.array(rules, dtype=np.uint8),
        path_paint=np.array(paints, dtype=np.float64),
        viewport=(0, 0, size, size),
    )


def make_tall_scene(size: int, n_paths: int, blocks: int):
    """Weak-scaling workload: `blocks` scenes of `n_paths` paths stacked vertically on a (blocks*size) x size canvas.
    Block b is make_scene(size, n_paths, seed=SEED + b) moved down by b*size rows; paths near a block border reach
    into the neighbour block, exactly like paths near a tile border of one big drawing.  Paint order = block, then i."""
    parts = [make_scene(size, n_paths, seed=SEED + b) for b in range(blocks)]
    segs = []
    for b, sc in enumerate(parts):
        s = sc["segs"].copy()
        s[:, 1::2] += float(b * size)  # y of every control point (row after the swap transform)
        segs.append(s)
    offs = [0]
    for sc in parts:
        offs.extend((sc["path_seg_off"][1:] + offs[-1]).tolist())
    segs = np.concatenate(segs)
    return dict(
        segs=segs,
        seg_kind=np.ones(len(segs), dtype=np.uint8),
        path_seg_off=np.array(offs, dtype=np.int64),
        path_m6=np.tile(SWAP_M6, (n_paths * blocks, 1)),
        path_rule=np.concatenate([sc["path_rule"] for sc in parts]),
        path_paint=np.concatenate([sc["path_paint"] for sc in parts]),
        viewport=(0, 0, size * blocks, size),
    )


def rows_subscene(scene, r0: int, r1: int):
    """The paths of `scene` whose control points reach the rows [r0, r1) (+-2 rows of slack), in paint order, with
    viewport = that row block: what one GPU of a row-sharded render needs (a path that crosses the border goes to both
    neighbours).  The curve stays inside the hull of its control points, so nothing that has a pixel in the block is lost."""
    off = scene["path_seg_off"]
    y = scene["segs"][:, 1::2]  # rows after the swap transform
    seg_lo, seg_hi = y.min(axis=1), y.max(axis=1)
    n = len(off) - 1
    lo = np.minimum.reduceat(seg_lo, off[:-1]) if n else np.zeros(0)
    hi = np.maximum.reduceat(seg_hi, off[:-1]) if n else np.zeros(0)
    keep = np.nonzero((np.floor(lo) - 2 < r1) & (np.ceil(hi) + 2 > r0))[0]
    seg_idx = np.concatenate([np.arange(off[p], off[p + 1]) for p in keep]) if len(keep) else np.zeros(0, dtype=np.int64)
    sizes = (off[1:] - off[:-1])[keep]
    return dict(
        segs=scene["segs"][seg_idx],
        seg_kind=scene["seg_kind"][seg_idx],
        path_seg_off=np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64),
        path_m6=scene["path_m6"][keep],
        path_rule=scene["path_rule"][keep],
        path_paint=scene["path_paint"][keep],
        viewport=(r0, scene["viewport"][1], r1 - r0, scene["viewport"][3]),
    ), keep


def presentation_segs(scene) -> np.ndarray:
    """The scene's control points after the swap transform, (n, 8) as (row, col) pairs.
    The swap matrix only permutes coordinates (exact in floating point)."""
    s = scene["segs"].reshape(-1, 4, 2)
    return np.ascontiguousarray(s[:, :, ::-1]).reshape(-1, 8)
