#!/usr/bin/env python3
"""Per-workgroup timeline of k_tile_render (diagnostic build -DSVGR_DBG_TIMELINE, file written by the library when
$SVGR_DBG_TIMELINE names a path): how full the chip is over the launch, what a workgroup's lifetime depends on, how long
the tail is.

    python profiles/timeline.py gpurun_out/timeline.bin [n_ctiles]
"""
import sys

import numpy as np


def main():
    allw = np.fromfile(sys.argv[1], dtype=np.uint64)[8:]
    phases = sub = None
    if allw.size >= 12 * 65536:  # (round 4: a third table with the parts of the tile switch)
        sub = allw[8 * 65536: 12 * 65536].reshape(-1, 4)
    if allw.size >= 8 * 65536:   # (newer builds: a second table with the phase stamps of every workgroup)
        phases = allw[4 * 65536: 8 * 65536].reshape(-1, 4)
        allw = allw[: 4 * 65536]
    raw = allw.reshape(-1, 4)
    live = raw[:, 1] > 0
    t = raw[live]
    if phases is not None:
        ph = phases[live].astype(np.int64)
        st_, en_, it_ = t[:, 0].astype(np.int64), t[:, 1].astype(np.int64), t[:, 2].astype(np.int64)
        has = (it_ > 0) & (ph[:, 3] > 0)
        if has.any():
            # (round 4: sums over a workgroup's tiles -- [0] end of a tile's items (launch start) -> the wait for the next tile's first
            #  loads, [1] that wait, [2] the item rounds; [3] tiles)
            tiles = ph[has, 3].astype(float)
            d0 = ph[has, 0] * 10e-3; d1 = ph[has, 1] * 10e-3; d2 = ph[has, 2] * 10e-3
            life_ = (en_[has] - st_[has]) * 10e-3
            if sub is not None:
                sb = sub[live][has].astype(float) * 10e-3
                print(f"  of the switch: registers -> float32 -> LDS {(sb[:, 0] / tiles).mean():.2f}, next tile's page taken + its first loads issued "
                      f"{(sb[:, 1] / tiles).mean():.2f}, rows LDS -> stores + re-zero {(sb[:, 2] / tiles).mean():.2f} (the rest: accumulators zeroed, loop exit)")
                wq = sub[live][has][:, 3]
                if wq.any():   # (-DSVGR_DBG_TL_WAITS: shader cycles of wave 0 in the loop's barrier / in the wait for the iteration's loads)
                    wb = (wq & np.uint64(0xFFFFFFFF)).astype(float); wi = (wq >> np.uint64(32)).astype(float)
                    n_it = it_[has].sum()
                    print(f"  waits of wave 0 per item (shader cycles, the two clock reads included): barrier {wb.sum() / n_it:.0f}, iteration's loads {wi.sum() / n_it:.0f}"
                          f"  (an item: {d2.sum() / n_it * 2400:.0f} cycles at 2.4 GHz)")
            print(f"phases per TILE (mean us; {int(has.sum())} workgroups, {tiles.mean():.2f} tiles each): switch (stores, next tile's loads issued) "
                  f"{(d0 / tiles).mean():.2f}, wait for the first loads {(d1 / tiles).mean():.2f}, item rounds {(d2 / tiles).mean():.2f} "
                  f"({(d2.sum() / it_[has].sum()):.3f} per item), rest of the workgroup's life (last store, exit) {((life_ - d0 - d1 - d2) / tiles).mean():.2f}")
    start, end, items = t[:, 0].astype(np.int64), t[:, 1].astype(np.int64), t[:, 2].astype(np.int64)
    hw, xcc = (t[:, 3] & np.uint64(0xFFFFFFFF)).astype(np.int64), (t[:, 3] >> np.uint64(32)).astype(np.int64) & 0xF
    t0 = start.min()
    start, end = (start - t0) * 10e-3, (end - t0) * 10e-3  # us (100 MHz clock)
    life = end - start
    span = end.max()
    print(f"workgroups {len(t)}, span {span:.1f} us, items/WG mean {items.mean():.1f} max {items.max()}, "
          f"lifetime mean {life.mean():.1f} us  p50 {np.percentile(life, 50):.1f}  p99 {np.percentile(life, 99):.1f}  max {life.max():.1f}")
    # lifetime ~ a + b * items
    A = np.stack([np.ones_like(items, dtype=float), items.astype(float)], 1)
    coef, *_ = np.linalg.lstsq(A, life, rcond=None)
    print(f"lifetime = {coef[0]:.2f} us + {coef[1]:.3f} us x items   (residual rms {np.sqrt(np.mean((A @ coef - life) ** 2)):.2f} us)")
    # occupancy over time
    edges = np.linspace(0, span, 41)
    print("time slice (us): resident workgroups (mean)")
    tot_slot_time = 0.0
    for a, b in zip(edges[:-1], edges[1:]):
        ov = np.clip(np.minimum(end, b) - np.maximum(start, a), 0, None).sum() / (b - a)
        tot_slot_time += ov * (b - a)
        print(f"  {a:7.1f}-{b:7.1f}: {ov:7.0f}")
    print(f"resident-workgroup time {tot_slot_time:.0f} WG.us = {tot_slot_time / span:.0f} mean resident")
    # last starts / tail
    print(f"last workgroup starts at {start.max():.1f} us; workgroups alive in the last 10% of the span: "
          f"{int((end > 0.9 * span).sum())}")
    # per XCC / CU balance
    cu = (hw >> 8) & 0xF
    se = (hw >> 13) & 0x7
    key = xcc * 1000 + se * 16 + cu
    uniq, cnt = np.unique(key, return_counts=True)
    busy = np.array([life[key == k].sum() for k in uniq])
    print(f"distinct (xcc, se, cu): {len(uniq)}; WGs per CU min {cnt.min()} max {cnt.max()}; busy WG.us per CU min {busy.min():.0f} "
          f"mean {busy.mean():.0f} max {busy.max():.0f}")
    for x in range(8):
        m = xcc == x
        if m.any():
            print(f"  xcc {x}: {int(m.sum())} WGs, items {int(items[m].sum())}, last end {end[m].max():.1f} us")
    if len(sys.argv) > 2:
        n_ct = int(sys.argv[2])
        idx = np.nonzero(live)[0]
        order = np.argsort(start)
        print("dispatch order (first 16 by start): wg ids", idx[order[:16]].tolist(), "n_ct", n_ct)


if __name__ == "__main__":
    main()
