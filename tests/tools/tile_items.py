"""Items per tile of the bench scene (CPU, the oracle's masks): a (path, tile) pair is an item of the tile kernel when the path has
coverage >= 1e-6 somewhere in the tile -- classes 1 and 2 of k_path_build.  The histogram VERDICT r3 #1d asks for beside the
timeline.   python tests/tools/tile_items.py [size] [paths]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
from oracle import oracle  # noqa: E402
from svgrasterize_amd import synth  # noqa: E402

size = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
n = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
TR, TC = 16, 64
sc = synth.make_scene(size, n)
segs = synth.presentation_segs(sc)
off = sc["path_seg_off"]
grid = np.zeros((size // TR, size // TC), dtype=np.int32)
for p in range(n):
    cub = segs[off[p]:off[p + 1]].reshape(-1, 4, 2)
    r = oracle.path_mask(np.zeros((0, 2, 2)), cub, None, "evenodd" if sc["path_rule"][p] else "nonzero", sc["viewport"])
    if r is None:
        continue
    m, (r0, c0), _ = r
    rows, cols = m.shape
    R0, C0 = (r0 // TR) * TR, (c0 // TC) * TC
    R1, C1 = -(-(r0 + rows) // TR) * TR, -(-(c0 + cols) // TC) * TC
    M = np.zeros((R1 - R0, C1 - C0), dtype=bool)
    M[r0 - R0:r0 - R0 + rows, c0 - C0:c0 - C0 + cols] = m >= 1e-6
    vis = M.reshape((R1 - R0) // TR, TR, (C1 - C0) // TC, TC).any(axis=(1, 3))
    grid[R0 // TR:R1 // TR, C0 // TC:C1 // TC] += vis
h = np.bincount(grid.ravel())
tot = grid.size
print(f"synthetic {n} paths @ {size} x {size}: {tot} tiles of {TR} x {TC}, {int(grid.sum())} items, mean {grid.mean():.2f}, max {grid.max()}")
acc = 0
for k, c in enumerate(h):
    acc += c
    print(f"  {k:3d} items: {c:6d} tiles ({100.0 * c / tot:5.2f} %, cumulative {100.0 * acc / tot:6.2f} %)  items in them {100.0 * k * c / max(grid.sum(), 1):5.2f} %")
