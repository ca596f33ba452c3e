#!/usr/bin/env python3
"""Experiment: one canvas as K row strips, each its own batch (rank r of K, one strip per rank) on its own HIP stream of the
SAME GPU, so that the latency-bound geometry kernels of one strip overlap the VALU-bound tile kernel of another.

    python profiles/overlap_strips.py [--k 1 2 4] [--steps 100] [--workload synth4096]
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
    ap.add_argument("--k", type=int, nargs="+", default=[1, 2, 4])
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--workload", default="synth4096")
    args = ap.parse_args()
    import bench
    from svgrasterize_amd import _abi

    sc, _ = bench.load_workload(args.workload)
    rows, cols = int(sc["viewport"][2]), int(sc["viewport"][3])
    n_bands = (rows + _abi.tile_rows() - 1) // _abi.tile_rows()
    for k in args.k:
        ctxs = [_abi.Context(0) for _ in range(k)]
        batches, outs = [], []
        for r, ctx in enumerate(ctxs):
            b = _abi.Batch(ctx, sc["segs"], sc["seg_kind"], sc["path_seg_off"], sc["path_m6"], sc["path_rule"], sc["path_paint"],
                           viewport=sc["viewport"])
            if k > 1:
                b.set_bands(r, k, (n_bands + k - 1) // k)
            b.plan()
            batches.append(b)
            outs.append(ctx.alloc(max(b.owned_rows(), 1) * cols * 16))

        def step():
            for b, o in zip(batches, outs):
                b.render(o, _abi.OUT_CANVAS_F32, _abi.RENDER_CLIP01)

        for _ in range(10):
            step()
        for c in ctxs:
            c.sync()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step()
        for c in ctxs:
            c.sync()
        dt = (time.perf_counter() - t0) / args.steps
        print(json.dumps({"strips": k, "ms_per_step": round(dt * 1e3, 4)}), flush=True)
        for b in batches:
            b.destroy()


if __name__ == "__main__":
    main()
