#!/usr/bin/env python3
"""bench.py -- AA coverage + composite throughput on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload synth4096|synth8192|tiger2048]

One step = one full pass of the hot path over the scene, inputs resident in HBM when the clock
starts: transform + flatten + bbox + band binning + tile kernel (LDS delta scatter, row scan, fill
rule, paint, source-over) -> finished float32 RGBA canvas in HBM.  No host read-back inside a step.

N > 1 (launched by torch.distributed.run, one rank per GPU).  The path shards by rows (a scanline never needs another
scanline), so the headline is WEAK scaling: a drawing N times as tall -- N blocks of the bench scene stacked, paths cross
the block borders -- and every GPU renders its own 4096 x 4096 block of rows from the paths whose control points reach
it (border paths go to both neighbours: duplicated geometry is the halo, no pixel ever crosses a GPU, no data-path
collective).  Per-GPU work is fixed; `value` = path-pixels of all blocks / slowest rank's time.  RCCL is used for the
barrier / max-over-ranks clock only.

The same line carries `strong_scaling`, measured right after: the ONE 4096 x 4096 bench scene sharded over the N GPUs by
interleaved strips of 128 scanlines (`svgr_batch_set_bands`; every rank culls the geometry to what reaches its strips),
total work fixed, plus the optional all_gather of the strips -- the north star's "tile-parallel speedup" of a
half-millisecond job, bounded by launch latencies.

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
    dist = torch = None
    coll_dev = "cpu"
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
            coll_dev = f"cuda:{local_rank}"
        else:
            dist.init_process_group(backend)
    if args.gpus != world and rank == 0 and world > 1:
        print(f"[bench] --gpus {args.gpus} but WORLD_SIZE={world}; using WORLD_SIZE", file=sys.stderr)

    import numpy as np  # noqa: F401

    import svgrasterize_amd as S
    from svgrasterize_amd import _abi, synth

    ctx = S.Context.get(local_rank)
    every = max(1, args.time_every)
    flags = _abi.RENDER_CLIP01

    def barrier():
        ctx.sync()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    def reduce(values, op):
        """all-reduce a list of floats over the ranks (identity for one rank)"""
        if dist is None:
            return list(values)
        t = torch.tensor(list(values), dtype=torch.float64, device=coll_dev)
        dist.all_reduce(t, op=op)
        return [float(v) for v in t.cpu()]

    def measure(batch, out):
        """W warm-up steps, barrier, exactly K timed steps, barrier -> (max-over-ranks seconds, stage timings of this rank)"""
        for _ in range(args.warmup):
            batch.render(out, _abi.OUT_CANVAS_F32, flags)
        barrier()
        batch.timings()  # drop
        t0 = time.perf_counter()
        for i in range(args.steps):
            batch.render(out, _abi.OUT_CANVAS_F32, flags | (_abi.RENDER_TIMED if i % every == 0 else 0))
        ctx.sync()
        t_local = time.perf_counter() - t0
        barrier()
        tm = batch.timings()  # HIP events on the library's stream around the stages of every `every`-th timed step
        t_max = reduce([t_local], dist.ReduceOp.MAX)[0] if dist is not None else t_local
        return t_max, tm

    def new_batch(scene):
        return _abi.Batch(ctx, scene["segs"], scene["seg_kind"], scene["path_seg_off"], scene["path_m6"], scene["path_rule"],
                          scene["path_paint"], viewport=scene["viewport"])

    sc, desc = load_workload(args.workload)
    rows, cols = int(sc["viewport"][2]), int(sc["viewport"][3])
    n_scene_paths = int(len(sc["path_seg_off"]) - 1)
    weak_mode = world > 1 and args.workload.startswith("synth")

    # ---- headline -------------------------------------------------------------------------------------------------
    if weak_mode:
        # a drawing `world` times as tall; this rank renders its own block of rows from the paths that reach it
        tall = synth.make_tall_scene(rows, n_scene_paths, world)
        mine, kept = synth.rows_subscene(tall, rank * rows, (rank + 1) * rows)
        batch = new_batch(mine)
        st = batch.plan()
        out = ctx.alloc(rows * cols * 16)
        t_max, tm = measure(batch, out)
        P_rank, E_rank = int(st.path_pixels), int(st.n_edges)
        P, E, inst = (int(v) for v in reduce([P_rank, E_rank, len(kept)], dist.ReduceOp.SUM))
        config = {
            "workload": f"{world} stacked blocks of: {desc}", "canvas": [rows * world, cols], "paths": n_scene_paths * world,
            "edges": E, "path_pixels": P,
            "sharding": f"{world} ranks, each renders its own {rows}-row block from the paths whose control points reach it "
                        f"({inst} path instances in all: border paths go to both neighbours); no data-path collective",
        }
        scaling = "weak"
        canvas_px = rows * world * cols
        del out
        batch.destroy()
    else:
        batch = new_batch(sc)
        st = batch.plan()
        P = P_rank = int(st.path_pixels)
        E = E_rank = int(st.n_edges)
        strip = int(os.environ.get("SVGR_STRIP_BANDS", str(max(1, 128 // _abi.tile_rows()))))  # 128 scanlines per strip
        if world > 1:  # a real-asset workload on several GPUs: one scene, row strips
            batch.set_bands(rank, world, strip)
            batch.plan()
            P_rank, E_rank = P / world, E / world
        out = ctx.alloc(max(batch.owned_rows(), 1) * cols * 16)
        t_max, tm = measure(batch, out)
        config = {
            "workload": desc, "canvas": [rows, cols], "paths": n_scene_paths, "edges": E, "path_pixels": P,
            "sharding": f"{world} ranks x interleaved strips of {strip} bands ({strip * _abi.tile_rows()} rows)" if world > 1 else "single GPU",
        }
        # (the N = 1 point of the synthetic series is the same scene as block 0 of the weak-scaling drawing)
        scaling = "weak" if args.workload.startswith("synth") and world == 1 else "strong"
        canvas_px = rows * cols

    # ---- companion for N > 1: the ONE bench scene sharded over the ranks (strong scaling) ---------------------------------
    strong = None
    if weak_mode:
        try:
            sb = new_batch(sc)
            st1 = sb.plan()
            strip = int(os.environ.get("SVGR_STRIP_BANDS", str(max(1, 128 // _abi.tile_rows()))))
            sb.set_bands(rank, world, strip)
            sb.plan()  # per-rank capacities: each rank keeps only the geometry that reaches its strips
            own_rows = sb.owned_rows()
            out_t = torch.empty((own_rows, cols, 4), dtype=torch.float32, device=f"cuda:{local_rank}")
            sout = ctx.wrap(out_t.data_ptr(), out_t.numel() * 4)
            s_max, _stm = measure(sb, sout)
            strong = {
                "scaling": "strong", "value": round(int(st1.path_pixels) / (s_max / args.steps) / 1e6, 1), "unit": "Mpixels/s",
                "ms_per_step": round(s_max / args.steps * 1e3, 4), "path_pixels": int(st1.path_pixels),
                "workload": f"the single-GPU bench scene ({n_scene_paths} paths @ {rows}x{cols}) sharded over {world} ranks by "
                            f"interleaved strips of {strip * _abi.tile_rows()} rows",
            }
            try:  # optional assembly of the full canvas on every rank (svgrasterize.py_amd/dist.py)
                from svgrasterize_amd import dist as sdist

                barrier()
                g0 = time.perf_counter()
                for _ in range(3):
                    full_t = sdist.gather_canvas(out_t if coll_dev != "cpu" else out_t.cpu(), rows, _abi.tile_rows(), strip=strip)
                torch.cuda.synchronize()
                strong["all_gather_ms"] = round((time.perf_counter() - g0) / 3 * 1e3, 4)
                del full_t
            except Exception as exc:  # noqa: BLE001
                print(f"[bench] rank {rank}: strip all_gather failed: {exc!r}", file=sys.stderr)
            del sout, out_t
            sb.destroy()
        except Exception as exc:  # noqa: BLE001  (a companion must not cost the bench line)
            print(f"[bench] rank {rank}: strong-scaling companion failed: {exc!r}", file=sys.stderr)
            strong = None

    if rank == 0:
        n_timed = max(tm["n"], 1)
        tile_ms = tm["ms_tile"] / n_timed
        geo_ms = tm["ms_geometry"] / n_timed
        # the dominant kernel as launched on this rank: its own share of the algorithmic bytes over its own duration
        alg_bytes = BYTES_PER_PATH_PIXEL * P_rank + BYTES_PER_EDGE * E_rank
        achieved = alg_bytes / (tile_ms * 1e-3) / 1e9 if tile_ms > 0 else 0.0
        traffic = None
        pmc = os.path.join(ROOT, "profiles", "pmc_tile_kernel.json")
        if os.path.exists(pmc) and world == 1:
            try:
                rec = json.load(open(pmc))
                if rec.get("workload") == args.workload:
                    traffic = rec.get("hbm_bytes_per_launch")
            except Exception:  # noqa: BLE001
                traffic = None
        line = {
            "metric": "Mpixels/sec AA coverage+composite (path-pixels/s; whole step: flatten+binning+coverage+composite)",
            "value": round(P / (t_max / args.steps) / 1e6, 1),
            "unit": "Mpixels/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(t_max / args.steps * 1e3, 4),
            "higher_is_better": True,
            "scaling": scaling,
            "vs_baseline": None,
            "dtype": "f64 arithmetic, f32 RGBA store",
            "data": "synthetic" if args.workload.startswith("synth") else "real asset (scene dump)",
            "config": config,
            "canvas_mpixels_per_s": round(canvas_px / (t_max / args.steps) / 1e6, 1),
            "roofline": {
                "kernel": "k_tile_render<f32>", "bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS,
                "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                "avg_launch_ms": round(tile_ms, 4), "geometry_ms": round(geo_ms, 4), "launches_timed": int(tm["n"]),
                "algorithmic_bytes_per_launch": int(alg_bytes),
                "note": "effective bandwidth: 40 B/path-pixel + 32 B/edge (SURVEY 8d) over the HIP-event duration of the "
                        "tile kernel (events on the library's stream around every %d-th launch of the timed region; rank 0's "
                        "launch and its share of the bytes); the kernel keeps trace and canvas on chip, so frac may exceed "
                        "what real HBM traffic could (`traffic` = measured HBM bytes per launch, rocprofv3 PMC); its own limit "
                        "is f64 VALU issue (about two thirds of the duration on synth4096, profiles/ + DESIGN.md section 4)" % every,
            },
        }
        if strong is not None:
            line["strong_scaling"] = strong
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(sc, args.cpu_paths)
        print(json.dumps(line))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
