#!/usr/bin/env python3
"""Experiment: consecutive renders of one drawing on K streams of the same GPU (K batches with their own work buffers, renders
dealt round-robin), so that the latency-bound geometry kernels of one render run beside the VALU-bound tile kernel of
another.  Measures renders per second, not the latency of one render.

    python profiles/overlap_renders.py [--k 1 2 3] [--steps 200] [--workload synth4096]
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
    ap.add_argument("--k", type=int, nargs="+", default=[1, 2, 3])
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--workload", default="synth4096")
    args = ap.parse_args()
    import bench
    from svgrasterize_amd import _abi

    sc, _ = bench.load_workload(args.workload)
    rows, cols = int(sc["viewport"][2]), int(sc["viewport"][3])
    for k in args.k:
        ctxs = [_abi.Context(0) for _ in range(k)]
        batches, outs = [], []
        for ctx in ctxs:
            b = _abi.Batch(ctx, sc["segs"], sc["seg_kind"], sc["path_seg_off"], sc["path_m6"], sc["path_rule"], sc["path_paint"],
                           viewport=sc["viewport"])
            b.plan()
            batches.append(b)
            outs.append(ctx.alloc(rows * cols * 16))
        for i in range(10 * k):
            batches[i % k].render(outs[i % k], _abi.OUT_CANVAS_F32, _abi.RENDER_CLIP01)
        for c in ctxs:
            c.sync()
        t0 = time.perf_counter()
        for i in range(args.steps):
            batches[i % k].render(outs[i % k], _abi.OUT_CANVAS_F32, _abi.RENDER_CLIP01)
        for c in ctxs:
            c.sync()
        dt = (time.perf_counter() - t0) / args.steps
        print(json.dumps({"streams": k, "ms_per_render": round(dt * 1e3, 4)}), flush=True)
        for b in batches:
            b.destroy()


if __name__ == "__main__":
    main()
