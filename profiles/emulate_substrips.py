#!/usr/bin/env python3
"""One rank's share of an N-way sharded render as K sub-strips, each a batch of its own on its own HIP stream of the SAME GPU:
the latency-chain geometry kernels of one sub-strip run beside the tile kernel of another (an eighth of config 4 leaves half
the chip idle).  Rank r of N with K sub-strips = sub-ranks K r .. K r + K - 1 of an N K-way sharding with strips 1 / K as tall.

    python profiles/emulate_substrips.py --world 8 --k 1 2 3 [--ranks 0 3 7] [--steps 60] [--workload synth8192]
"""
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
    ap.add_argument("--k", type=int, nargs="+", default=[1, 2, 3])
    ap.add_argument("--ranks", type=int, nargs="+", default=None)
    ap.add_argument("--steps", type=int, default=60)
    ap.add_argument("--workload", default="synth8192")
    args = ap.parse_args()
    import bench
    from svgrasterize_amd import _abi, dist as sdist

    sc, _ = bench.load_workload(args.workload)
    rows, cols = int(sc["viewport"][2]), int(sc["viewport"][3])
    strip = sdist.default_strip_bands(rows, _abi.tile_rows(), args.world)
    ranks = args.ranks if args.ranks is not None else list(range(args.world))
    for k in args.k:
        if strip % k:
            print(json.dumps({"sub_strips": k, "skipped": f"{strip} bands per strip do not split into {k}"}))
            continue
        ctxs = [_abi.Context(0) for _ in range(k)]
        per_rank = []
        for rank in ranks:
            batches, outs = [], []
            for j, ctx in enumerate(ctxs):
                b = _abi.Batch(ctx, sc["segs"], sc["seg_kind"], sc["path_seg_off"], sc["path_m6"], sc["path_rule"], sc["path_paint"],
                               viewport=sc["viewport"])
                b.set_bands(rank * k + j, args.world * k, strip // k)
                b.plan()
                batches.append(b)
                outs.append(ctx.alloc(max(b.owned_rows(), 1) * cols * 16))

            def step():
                for b, o in zip(batches, outs):
                    b.render(o, _abi.OUT_CANVAS_F32, _abi.RENDER_CLIP01)

            for _ in range(5):
                step()
            for c in ctxs:
                c.sync()
            t0 = time.perf_counter()
            for _ in range(args.steps):
                step()
            for c in ctxs:
                c.sync()
            per_rank.append(round((time.perf_counter() - t0) / args.steps * 1e3, 4))
            for b in batches:
                b.destroy()
        print(json.dumps({"world": args.world, "sub_strips": k, "ranks": ranks, "ms_step": per_rank, "slowest": max(per_rank)}), flush=True)


if __name__ == "__main__":
    main()
