"""Offline statistics of the bench scene's cells (CPU, oracle masks): what fraction of the tile kernel's item-waves could be
skipped or take a cheaper path if a cell carried a class per 8-row half.  tests/tools/cell_stats.py [size] [paths]"""
import sys, os
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
from oracle import oracle
from svgrasterize_amd import synth

size = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
n = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
TR, TC = 16, 64
sc = synth.make_scene(size, n)
segs = synth.presentation_segs(sc)
off = sc["path_seg_off"]
from collections import Counter
cls_cnt = Counter(); half_cnt = Counter(); c1_sat = Counter(); slot_vis = Counter(); px = Counter()
q_cnt = Counter()
for p in range(n):
    cub = segs[off[p]:off[p + 1]].reshape(-1, 4, 2)
    r = oracle.path_mask(np.zeros((0, 2, 2)), cub, None, "evenodd" if sc["path_rule"][p] else "nonzero", sc["viewport"])
    if r is None: continue
    m, (r0, c0), _ = r
    rows, cols = m.shape
    # embed into tile-aligned array
    R0 = (r0 // TR) * TR; C0 = (c0 // TC) * TC
    R1 = -(-(r0 + rows) // TR) * TR; C1 = -(-(c0 + cols) // TC) * TC
    M = np.zeros((R1 - R0, C1 - C0)); inl = np.zeros_like(M, dtype=bool)
    M[r0 - R0:r0 - R0 + rows, c0 - C0:c0 - C0 + cols] = m
    inl[r0 - R0:r0 - R0 + rows, c0 - C0:c0 - C0 + cols] = True
    nb, nc = M.shape[0] // TR, M.shape[1] // TC
    T = M.reshape(nb, TR, nc, TC).transpose(0, 2, 1, 3)        # (band, ct, 16, 64)
    I = inl.reshape(nb, TR, nc, TC).transpose(0, 2, 1, 3)
    vis = T >= 1e-6
    # a row is "touched" when the mask is not constant over the layer's columns inside the tile
    Tmax = np.where(I, T, -1).max(axis=3); Tmin = np.where(I, T, 2).min(axis=3)
    touched = (Tmax != Tmin) & I.any(axis=3)                    # (band, ct, 16)
    rowvis = vis.any(axis=3)
    for b in range(nb):
        for c in range(nc):
            t = touched[b, c]; rv = rowvis[b, c]
            if t.any(): cls = 2
            elif rv.any(): cls = 1
            else: cls = 0
            cls_cnt[cls] += 1
            if cls == 0: continue
            px[(cls, "vis")] += int(vis[b, c].sum()); px[(cls, "all")] += TR * TC
            for h in range(2):
                th = t[8 * h:8 * h + 8]; rh = rv[8 * h:8 * h + 8]
                hc = 2 if th.any() else (1 if rh.any() else 0)
                half_cnt[(cls, hc)] += 1
                # slots: pixel i of each 8-px chunk over the half's 8 rows
                v = vis[b, c, 8 * h:8 * h + 8].reshape(8, 8, 8)   # row, chunk, i
                anyslot = v.any(axis=(0, 1))
                slot_vis[(cls, hc)] += int(anyslot.sum())
                # compact 8x8 blocks (chunk-wise) for comparison
                slot_vis[(cls, hc, "blk")] += int(v.any(axis=(0, 2)).sum())
                if hc == 1:
                    sat = bool((T[b, c, 8 * h:8 * h + 8][rh] >= 1.0).all()) and bool(I[b, c, 8 * h:8 * h + 8][rh].all())
                    c1_sat[(cls, sat, int(rh.all()))] += 1
            for q in range(4):
                tq = t[4 * q:4 * q + 4]; rq = rv[4 * q:4 * q + 4]
                q_cnt[(cls, 2 if tq.any() else (1 if rq.any() else 0))] += 1
print("cells by class", dict(cls_cnt))
print("halves (cell class, half class)", dict(sorted(half_cnt.items())))
print("quarters (cell class, quarter class)", dict(sorted(q_cnt.items())))
print("class-1 halves (cell class, saturated and inside, all rows visible)", dict(sorted(c1_sat.items())))
print("visible px", dict(px))
print("slots with a visible pixel (of 8 per half)", {k: v for k, v in sorted(slot_vis.items(), key=str)})
