#!/usr/bin/env python3
"""bench.py -- AA coverage + composite throughput on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload synth4096|synth8192|tiger2048]

One step = one full pass of the hot path over the scene, inputs resident in HBM when the clock
starts: transform + flatten + bbox + band binning + tile kernel (LDS delta scatter, row scan, fill
rule, paint, source-over) -> finished float32 RGBA canvas in HBM.  No host read-back inside a step.

N > 1 (launched by torch.distributed.run, one rank per GPU): the canvas is sharded by interleaved
strips of 128 scanlines (rank r owns strips r, r+N, ...): the same scene, total work fixed -> "strong" scaling.
Edges that cross a band border are simply processed by both owners (duplicated edges are the halo;
no pixel ever crosses a GPU), so the data path needs no collective; RCCL is used for the barrier /
max-over-ranks clock and for the optional final all_gather of the bands (reported separately).

For N > 1 the line also carries a `weak_scaling` object, measured right after the strong-scaling steps: a drawing N
times as tall (one block of the bench scene per GPU, stacked; paths cross the block borders), every GPU rendering its own
4096-row block from the paths that reach it -- fixed work per GPU.  It never replaces `value`.

Rank 0 prints ONE JSON line (contract in the task statement + `roofline` and `cpu_baseline`).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X spec (guides/MI355X_MICROARCH.md: 8.0 TB/s; ~6.3 TB/s achievable copy)
BYTES_PER_PATH_PIXEL = 40  # read f64 trace 8 + read f32 RGBA 16 + write f32 RGBA 16
BYTES_PER_EDGE = 32        # 4 doubles, read once


def load_workload(name: str):
    import numpy as np

    from svgrasterize_amd import synth

    if name.startswith("synth"):
        size = int(name[5:])
        n = {4096: 4096, 8192: 10000}.get(size, size)
        sc = synth.make_scene(size, n)
        desc = f"synthetic {n} random closed cubic paths @ {size}x{size} (SURVEY 8d generator, splitmix64 0x5F3759DF)"
        return sc, desc
    if name == "tiger2048":
        import svgrasterize_amd as S
        from svgrasterize_amd import scenedump

        scene, info, _ = scenedump.load_scene(os.path.join(ROOT, "tests", "golden", "scene_tiger.npz"))
        tr = S.Transform().matrix(0, 1, 0, 1, 0, 0)
        leaves = scene.leaves(tr, linear_rgb=False)
        segs, kinds, offs = [], [], [0]
        for path, _m6, _rule, _paint, _flags in leaves:
            s, k = path.packed()
            segs.append(s)
            kinds.append(k)
            offs.append(offs[-1] + len(s))
        h, w = info["size"]
        sc = dict(segs=np.concatenate(segs), seg_kind=np.concatenate(kinds), path_seg_off=np.array(offs, dtype=np.int64),
                  path_m6=np.array([l[1] for l in leaves]), path_rule=np.array([l[2] | (l[4] << 1) for l in leaves], dtype=np.uint8),
                  path_paint=np.array([l[3] for l in leaves]), viewport=(0, 0, h, w))
        return sc, "Ghostscript tiger (scene dump, 182 solid fills incl. pre-stroked outlines) @ 2048x2048"
    raise SystemExit(f"unknown workload {name}")


def cpu_baseline(sc, budget_paths: int | None = None):
    """The CPU oracle (C restatement of the reference passes, oracle/svgr_oracle.c) timed on this
    host, single thread, on the same scene (or its first `budget_paths` paths)."""
    import ctypes as C

    import numpy as np

    from oracle import oracle as orc

    n_all = len(sc["path_seg_off"]) - 1
    n = n_all if budget_paths is None else min(budget_paths, n_all)
    off = np.ascontiguousarray(sc["path_seg_off"][: n + 1])
    m6 = sc["path_m6"]
    segs = sc["segs"][: off[-1]]
    seg_path = np.repeat(np.arange(n), np.diff(off))
    pts = segs.reshape(-1, 4, 2)
    pres = np.empty_like(pts)
    for p in np.unique(seg_path):  # transform on the host, outside the timed region
        sel = seg_path == p
        m = np.eye(3)
        m[:2, :] = m6[p].reshape(2, 3)
        pres[sel] = orc.transform_points(m, pts[sel])
    pres = np.ascontiguousarray(pres.reshape(-1, 8))
    vp = np.array(sc["viewport"], dtype=np.int64)
    canvas = np.zeros((int(vp[2]), int(vp[3]), 4))
    canvas.fill(0.0)  # pre-fault: page faults are not the algorithm
    stats = np.zeros(2, dtype=np.int64)
    L = orc.lib()
    args = (pres.reshape(-1), np.ascontiguousarray(sc["seg_kind"][: off[-1]]), off, n,
            np.ascontiguousarray(sc["path_rule"][:n]), np.ascontiguousarray(sc["path_paint"][:n]).reshape(-1), vp, 1,
            canvas.reshape(-1), stats.ctypes.data_as(C.c_void_p))
    t0 = time.perf_counter()
    rc = L.orc_render_solid(*args)
    dt = time.perf_counter() - t0
    if rc != 0:
        raise RuntimeError(f"oracle failed: {rc}")
    P1 = int(stats[0])
    # the same render cut into row strips, one per thread, on this process's share of the host cores
    threads = max(1, min(16, len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)))
    canvas.fill(0.0)
    t0 = time.perf_counter()
    rc = L.orc_render_solid_strips(*args, threads, threads)
    dt_mt = time.perf_counter() - t0
    if rc != 0:
        raise RuntimeError(f"oracle (strips) failed: {rc}")
    return dict(
        value=round(P1 / dt / 1e6, 3), unit="Mpixels/s (path-pixels)", cores=1, kind="port",
        sample=f"first {n} of {n_all} paths of the same scene, full viewport, {dt:.2f} s, P={P1} "
               f"(oracle/svgr_oracle.c: pass-by-pass C restatement of the reference, float64, 1 thread of {os.cpu_count()})",
        all_cores_value=round(int(stats[0]) / dt_mt / 1e6, 3), all_cores=threads,
        all_cores_sample=f"same render as {threads} row strips (the reference's viewport cropping) on {threads} OpenMP threads, {dt_mt:.2f} s",
    )


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--workload", default="synth4096")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-paths", type=int, default=None, help="limit the CPU baseline to the first N paths")
    ap.add_argument("--time-every", type=int, default=4,
                    help="bracket the stages of every n-th timed step with HIP events (each event drains the queue for ~8 us, "
                         "so timing every step would slow the thing being measured)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    if world > 1:
        # torch first: its bundled HIP runtime must be the one in the process
        import torch
        import torch.distributed as dist

        # rehearsal on a one-GPU box: SVGR_BENCH_BACKEND=gloo SVGR_BENCH_DEVICE=0 puts every rank on GPU 0 and
        # runs the collectives on CPU tensors (RCCL refuses two ranks on one device)
        backend = os.environ.get("SVGR_BENCH_BACKEND", "nccl")
        if "SVGR_BENCH_DEVICE" in os.environ:
            local_rank = int(os.environ["SVGR_BENCH_DEVICE"])
        torch.cuda.set_device(local_rank)
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)
    if args.gpus != world and rank == 0 and world > 1:
        print(f"[bench] --gpus {args.gpus} but WORLD_SIZE={world}; using WORLD_SIZE", file=sys.stderr)

    import numpy as np

    import svgrasterize_amd as S
    from svgrasterize_amd import _abi

    ctx = S.Context.get(local_rank)
    sc, desc = load_workload(args.workload)
    rows, cols = int(sc["viewport"][2]), int(sc["viewport"][3])
    batch = _abi.Batch(ctx, sc["segs"], sc["seg_kind"], sc["path_seg_off"], sc["path_m6"], sc["path_rule"], sc["path_paint"],
                       viewport=sc["viewport"])
    st = batch.plan()
    P, E = int(st.path_pixels), int(st.n_edges)
    strip = int(os.environ.get("SVGR_STRIP_BANDS", str(max(1, 128 // _abi.tile_rows()))))  # 128 scanlines per strip
    if world > 1:
        batch.set_bands(rank, world, strip)
        batch.plan()  # per-rank capacities: each rank keeps only the edges that reach its strips
    own_rows = batch.owned_rows()
    if world > 1:
        import torch

        out_t = torch.empty((own_rows, cols, 4), dtype=torch.float32, device=f"cuda:{local_rank}")
        out = ctx.wrap(out_t.data_ptr(), out_t.numel() * 4)
    else:
        out = ctx.alloc(own_rows * cols * 16)

    def barrier():
        ctx.sync()
        if dist is not None:
            import torch

            dist.barrier()
            torch.cuda.synchronize()

    flags = _abi.RENDER_CLIP01
    for _ in range(args.warmup):
        batch.render(out, _abi.OUT_CANVAS_F32, flags)
    barrier()
    batch.timings()  # drop
    t0 = time.perf_counter()
    every = max(1, args.time_every)
    for i in range(args.steps):
        batch.render(out, _abi.OUT_CANVAS_F32, flags | (_abi.RENDER_TIMED if i % every == 0 else 0))
    ctx.sync()
    t_local = time.perf_counter() - t0
    barrier()
    tm = batch.timings()  # HIP events on the library's stream around the stages of every timed step

    t_max = t_local
    gather_ms = None
    if dist is not None:
        import torch

        coll_dev = f"cuda:{local_rank}" if dist.get_backend() == "nccl" else "cpu"
        tt = torch.tensor([t_local], dtype=torch.float64, device=coll_dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        t_max = float(tt.item())
        # optional assembly of the full canvas on every rank (svgrasterize.py_amd/dist.py), reported separately
        from svgrasterize_amd import dist as sdist

        try:  # (reported separately; a failure here must not cost the bench line)
            dist.barrier()
            torch.cuda.synchronize()
            g0 = time.perf_counter()
            for _ in range(3):
                full_t = sdist.gather_canvas(out_t if coll_dev != "cpu" else out_t.cpu(), rows, _abi.tile_rows(), strip=strip)
            torch.cuda.synchronize()
            gather_ms = (time.perf_counter() - g0) / 3 * 1e3
            del full_t
        except Exception as exc:  # noqa: BLE001
            print(f"[bench] rank {rank}: strip all_gather failed: {exc!r}", file=sys.stderr)
            gather_ms = None

    # Weak-scaling companion (world > 1, synthetic workloads): a drawing `world` times as tall (one block of the bench
    # scene per GPU, stacked; paths cross the block borders), each GPU renders its own block of rows from the paths that
    # reach it.  Per-GPU work stays fixed as N grows; reported next to the strong-scaling value, never instead of it.
    weak = None
    if dist is not None and args.workload.startswith("synth"):
        try:
            from svgrasterize_amd import synth

            size = int(args.workload[5:])
            n_block = {4096: 4096, 8192: 10000}.get(size, size)
            tall = synth.make_tall_scene(size, n_block, world)
            sub, kept = synth.rows_subscene(tall, rank * size, (rank + 1) * size)
            wb = _abi.Batch(ctx, sub["segs"], sub["seg_kind"], sub["path_seg_off"], sub["path_m6"], sub["path_rule"],
                            sub["path_paint"], viewport=sub["viewport"])
            wst = wb.plan()
            wout = ctx.alloc(size * cols * 16)
            for _ in range(args.warmup):
                wb.render(wout, _abi.OUT_CANVAS_F32, flags)
            barrier()
            w0 = time.perf_counter()
            for _ in range(args.steps):
                wb.render(wout, _abi.OUT_CANVAS_F32, flags)
            ctx.sync()
            w_local = time.perf_counter() - w0
            barrier()
            acc = torch.tensor([w_local, float(wst.path_pixels), float(len(kept))], dtype=torch.float64, device=coll_dev)
            tmax_t = acc[:1].clone()
            dist.all_reduce(tmax_t, op=dist.ReduceOp.MAX)
            dist.all_reduce(acc, op=dist.ReduceOp.SUM)
            w_t = float(tmax_t.item())
            weak = {
                "scaling": "weak", "value": round(float(acc[1].item()) / (w_t / args.steps) / 1e6, 1), "unit": "Mpixels/s",
                "ms_per_step": round(w_t / args.steps * 1e3, 4), "path_pixels": int(acc[1].item()),
                "workload": f"{world} stacked blocks of the bench scene ({world * n_block} paths @ {world * size}x{size}); "
                            f"every GPU renders its own {size}-row block from the paths that reach it "
                            f"({int(acc[2].item())} path instances over all GPUs: border paths go to both neighbours)",
            }
            del wout
            wb.destroy()
        except Exception as exc:  # noqa: BLE001
            print(f"[bench] rank {rank}: weak-scaling companion failed: {exc!r}", file=sys.stderr)
            weak = None

    if rank == 0:
        ms_step = t_max / args.steps * 1e3
        tile_ms = tm["ms_tile"] / max(tm["n"], 1)
        geo_ms = tm["ms_geometry"] / max(tm["n"], 1)
        # per-rank share of the algorithmic bytes: this rank's tile kernel handled ~1/world of the path-pixels
        alg_bytes = (BYTES_PER_PATH_PIXEL * P + BYTES_PER_EDGE * E) / world
        achieved = alg_bytes / (tile_ms * 1e-3) / 1e9 if tile_ms > 0 else 0.0
        traffic = None
        pmc = os.path.join(ROOT, "profiles", "pmc_tile_kernel.json")
        if os.path.exists(pmc) and world == 1:
            try:
                rec = json.load(open(pmc))
                if rec.get("workload") == args.workload:
                    traffic = rec.get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        line = {
            "metric": "Mpixels/sec AA coverage+composite (path-pixels/s; whole step: flatten+binning+coverage+composite)",
            "value": round(P / (t_max / args.steps) / 1e6, 1),
            "unit": "Mpixels/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(ms_step, 4),
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f64 arithmetic, f32 RGBA store",
            "data": "synthetic" if args.workload.startswith("synth") else "real asset (scene dump)",
            "config": {
                "workload": desc, "canvas": [rows, cols], "paths": int(len(sc["path_seg_off"]) - 1), "edges": E,
                "path_pixels": P, "sharding": f"{world} ranks x interleaved strips of {strip} bands ({strip * _abi.tile_rows()} rows)" if world > 1 else "single GPU",
            },
            "canvas_mpixels_per_s": round(rows * cols / (t_max / args.steps) / 1e6, 1),
            "roofline": {
                "kernel": "k_tile_render<f32>", "bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS,
                "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                "avg_launch_ms": round(tile_ms, 4), "geometry_ms": round(geo_ms, 4), "launches_timed": int(tm["n"]),
                "algorithmic_bytes_per_launch": int(alg_bytes),
                "note": "effective bandwidth: 40 B/path-pixel + 32 B/edge (SURVEY 8d) over the HIP-event duration of the "
                        "tile kernel (events on the library's stream around every %d-th launch of the timed region); the kernel "
                        "keeps trace and canvas on chip, so frac may exceed what real HBM traffic could" % every,
            },
        }
        if gather_ms is not None:
            line["all_gather_ms"] = round(gather_ms, 4)
        if weak is not None:
            line["weak_scaling"] = weak
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(sc, args.cpu_paths)
        print(json.dumps(line))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
