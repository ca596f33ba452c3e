#!/usr/bin/env python3
"""Time ONE rank's share of an N-way band-sharded render on a single GPU (no process group).

    python profiles/emulate_rank.py --world 8 [--rank 0] [--strip 16] [--steps 50] [--workload synth4096]

The data path of the multi-GPU bench has no collective, so the per-rank step time measured here is what
`bench.py --gpus N` sees on rank `--rank` (up to launch jitter); `--all` loops over every rank and prints the
slowest one, i.e. the strong-scaling step time.  Used under rocprofv3 --kernel-trace to see which kernels
do not shrink with 1/N.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--world", type=int, default=8)
    ap.add_argument("--rank", type=int, default=0)
    ap.add_argument("--all", action="store_true")
    ap.add_argument("--strip", type=int, default=int(os.environ.get("SVGR_STRIP_BANDS", "0")), help="bands per strip (0: bench.py's default)")
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--workload", default="synth4096")
    ap.add_argument("--timed", action="store_true")
    ap.add_argument("--reverse", action="store_true", help="with --all: the last rank first (is the first one measured slower, or rank 0?)")
    args = ap.parse_args()

    import bench
    import svgrasterize_amd as S
    from svgrasterize_amd import _abi

    ctx = S.Context.get(0)
    sc, _ = bench.load_workload(args.workload)
    if args.strip <= 0:
        from svgrasterize_amd import dist as sdist

        args.strip = sdist.default_strip_bands(int(sc["viewport"][2]), _abi.tile_rows(), args.world)
    cols = int(sc["viewport"][3])
    batch = _abi.Batch(ctx, sc["segs"], sc["seg_kind"], sc["path_seg_off"], sc["path_m6"], sc["path_rule"], sc["path_paint"],
                       viewport=sc["viewport"])
    res = []
    order = list(range(args.world)) if args.all else [args.rank]
    if args.reverse:
        order.reverse()
    for rank in order:
        batch.set_bands(rank, args.world, args.strip)
        st = batch.plan()
        out = ctx.alloc(max(batch.owned_rows(), 1) * cols * 16)
        for _ in range(3):
            batch.render(out, _abi.OUT_CANVAS_F32, _abi.RENDER_CLIP01)
        ctx.sync()
        batch.timings()
        t0 = time.perf_counter()
        fl = _abi.RENDER_CLIP01 | (_abi.RENDER_TIMED if args.timed else 0)
        for _ in range(args.steps):
            batch.render(out, _abi.OUT_CANVAS_F32, fl)
        ctx.sync()
        dt = (time.perf_counter() - t0) / args.steps * 1e3
        tm = batch.timings()
        if not args.timed:
            tm = dict(n=1, ms_geometry=0.0, ms_tile=0.0)
        res.append(dict(rank=rank, ms_step=round(dt, 4), ms_geometry=round(tm["ms_geometry"] / tm["n"], 4),
                        ms_tile=round(tm["ms_tile"] / tm["n"], 4), edges=int(st.n_edges), records=int(st.n_band_segs)))
        del out
    worst = max(res, key=lambda r: r["ms_step"])
    print(json.dumps(dict(world=args.world, strip_bands=args.strip, workload=args.workload, slowest=worst, ranks=res)))


if __name__ == "__main__":
    main()
