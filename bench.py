#!/usr/bin/env python3
"""bench.py -- AA coverage + composite throughput on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload synth4096|synth8192|tiger2048]

One step = one full pass of the hot path over the scene, inputs resident in HBM when the clock
starts: transform + flatten + bbox + band binning + cell classification + tile kernel (LDS delta scatter,
row scan, fill rule, paint, source-over) -> finished float32 RGBA canvas in HBM.  No host read-back inside a step.

N = 1: the bench scene (BASELINE.json's metric configuration: 4096 paths @ 4096 x 4096).

N > 1 (launched by torch.distributed.run, one rank per GPU -- or, without a launcher, by this file itself: launch_ranks): STRONG scaling of BASELINE.json's config 4 -- the ONE
synthetic 10 000-path drawing @ 8192 x 8192, sharded over the N GPUs by strips of scanlines (one per GPU by default)
(`svgr_batch_set_bands`; every rank culls the geometry to what reaches its strips; edges that cross a strip border are
simply kept by both owners: duplicated geometry is the halo, no pixel crosses a GPU, no data-path collective).  Total work is
fixed; `value` = the drawing's path-pixels / the slowest rank's time.  RCCL carries the barrier / max-over-ranks clock and
the optional all_gather of the strips (`all_gather_ms`, reported beside the step, never inside it).  The same line carries
`single_gpu_same_scene` (rank 0 alone renders the whole drawing while the others wait: the denominator of the
tile-parallel speed-up, measured in the same run) and `weak_scaling` (N stacked 4096-row blocks, one per GPU).

Rank 0 prints ONE JSON line (contract in the task statement + `roofline` and `cpu_baseline`).

`roofline`: the dominant kernel is k_tile_render, timed live by HIP events on the library's stream.  It keeps the delta
tile and the canvas tile on chip, so what has to cross HBM per launch is the finished canvas once plus the geometry it reads:
  frac / achieved / peak   ALGORITHMIC floor over the launch time: max(t_hbm, t_f64) / t, where
                           t_hbm = (canvas bytes + 32 B per edge) / 8 TB/s and
                           t_f64 = (8 fma + 1 add) x visible path-pixels / 64 lanes x measured cycles per f64 wave-instruction
                                   / (1024 SIMDs x clock)   (profiles/valu_issue_mi355x.json: profiles/valu_issue.hip on this chip;
                                   the visible path-pixels come from the CPU oracle's census in the cpu_baseline leg)
                           `bound` names the larger of the two; with "hbm" achieved = floor bytes / t in GB/s against 8000
  traffic / hbm            measured HBM bytes per launch (2 x FETCH_SIZE + WRITE_SIZE, separate rocprofv3 --pmc passes)
  issue_utilisation        how busy the VALU issue slots were: the executed instruction mix (counters) priced with the measured
                           cycles per wave-instruction -- a utilisation, not a roofline fraction (executed, not algorithmic)
  effective_gbs            SURVEY 8d's algorithmic 40 B/path-pixel + 32 B/edge over t: a throughput figure, NOT a roofline
                           fraction (the reference's memory passes are what it prices; this kernel never makes them)
Counter values come from the committed profiles/pmc_kernels_<workload>.json (profiles/collect2.sh, separate --pmc passes
on the same command); `counters` names the file, the commit and whether the kernel source has changed since (`stale`).

`parity` (N = 1, with the cpu_baseline leg): the canvas of the LAST timed step, downloaded after the clock has stopped, against
the CPU oracle's canvas of the same scene under the float32 contract |got - f32(ref)| <= max(1 ULP, 2^-24): values, how many
are outside it, largest error.
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
N_SIMD = 1024          # 256 CUs x 4 SIMDs
CLOCK_GHZ = 2.4        # shader clock under load (profiles/valu_issue.hip reads 2405 s_memtime ticks per microsecond)
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
        for leaf in leaves:
            s, k = leaf[0].packed()
            segs.append(s)
            kinds.append(k)
            offs.append(offs[-1] + len(s))
        h, w = info["size"]
        sc = dict(segs=np.concatenate(segs), seg_kind=np.concatenate(kinds), path_seg_off=np.array(offs, dtype=np.int64),
                  path_m6=np.array([l[1] for l in leaves]), path_rule=np.array([l[2] | (l[4] << 1) for l in leaves], dtype=np.uint8),
                  path_paint=np.array([l[3] for l in leaves]), viewport=(0, 0, h, w))
        return sc, "Ghostscript tiger (scene dump, 182 solid fills incl. pre-stroked outlines) @ 2048x2048"
    raise SystemExit(f"unknown workload {name}")


SCENE_WORKLOADS = {  # real-asset configurations rendered through Scene.render (batched runs + per-node route)
    "icons4096": ("scene_icons4096.npz", "demo/icons.svg @4096x1051 (config 5: 431 gradient fills, 36 blurs up to 73x73, 123 opacity groups)"),
    "material4096": ("scene_material.npz", "demo/material-design.svg @4096x4096 (config 3: 989 fills, 935 clips)"),
}


def bench_scene(args):
    """`--workload icons4096 | material4096`: one step = one Scene.render of the document (host tree walk + every device
    launch it issues), result resident in HBM.  These configurations are bound by the host-side walk and per-node launch
    latencies, not by a kernel: the line carries the wall clock and canvas pixels/s; the per-kernel durations and HBM
    counters of the same command are in profiles/ (kernel_stats_<workload>*.csv, pmc_kernels_<workload>.json)."""
    # (the cyclic collector pauses while a top-level Scene.render runs: opt-in, asked for here -- read when the package is imported)
    os.environ.setdefault("SVGR_PAUSE_GC", "1")
    import numpy as np

    import svgrasterize_amd as S
    from svgrasterize_amd import scenedump

    fname, desc = SCENE_WORKLOADS[args.workload]
    S.set_render_cache(4)   # (opt-in: `ms_per_step` below is the re-render of an unchanged document; `cold_ms` what the default pays)
    ctx = S.Context.get(int(os.environ.get("LOCAL_RANK", "0")))
    scene, info, pins = scenedump.load_scene(os.path.join(ROOT, "tests", "golden", fname))
    h, w = info["size"]
    tr = S.Transform().matrix(0, 1, 0, 1, 0, 0)

    def step():
        layer, _hull = scene.render(tr, viewport=[0, 0, h, w], linear_rgb=False)
        return layer._device()

    # the FIRST render of the document in this process (leaf analysis, batch building, plans: what Scene.render retains for the
    # renders that follow), then the same with the retained state dropped each time: what a caller who renders a document once pays
    S.clear_render_cache()
    ctx.sync()
    c0 = time.perf_counter()
    step()
    ctx.sync()
    first_ms = (time.perf_counter() - c0) * 1e3
    cold, cold_listing = [], []
    for _ in range(3):   # (with the opt-in cache on: the render also lists the document's paint arrays for the cache's guard)
        S.clear_render_cache()
        ctx.sync()
        c0 = time.perf_counter()
        step()
        ctx.sync()
        cold_listing.append((time.perf_counter() - c0) * 1e3)
    S.set_render_cache(0)   # (the default: nothing kept between renders)
    for _ in range(3):
        ctx.sync()
        c0 = time.perf_counter()
        step()
        ctx.sync()
        cold.append((time.perf_counter() - c0) * 1e3)
    S.set_render_cache(4)
    for _ in range(args.warmup):
        step()
    ctx.sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        keep = step()
    ctx.sync()
    dt = (time.perf_counter() - t0) / args.steps
    del keep
    # the picture of a warm render against the reference's own (sparse pins of its full-size float32 canvas, made by
    # oracle/gen_golden.py --full from the imported reference): after the clock has stopped
    parity = None
    try:
        if "full_idx" in pins and "full_val" in pins:
            layer, _hull = scene.render(tr, viewport=[0, 0, h, w], linear_rgb=False)
            got = layer.to_canvas_f32(h, w).reshape(-1, 4)[pins["full_idx"]].astype(np.float64)
            ref32 = np.asarray(pins["full_val"], dtype=np.float32)
            a = np.abs(ref32)
            ulp = np.maximum((np.nextafter(a, np.float32(np.inf)) - a).astype(np.float64), 2.0 ** -24)
            err = np.abs(got - ref32.astype(np.float64))
            parity = {"values": int(err.size), "bad": int((err > ulp).sum()), "max_err": float(err.max(initial=0.0)),
                      "contract": "|got - f32(ref)| <= max(1 ULP_f32(ref), 2^-24)",
                      "what": "pins of the REFERENCE's own render of this document at this size (tests/golden, sha256 of its canvas "
                              f"{str(info.get('full', {}).get('sha256_f32', ''))[:16]}), checked on a warm render after the clock stopped"}
    except Exception as exc:  # noqa: BLE001
        parity = {"error": repr(exc)}
    counters, counters_file = load_counters(args.workload)
    stream = None
    if counters is not None:  # the streaming kernels of the per-node route: measured bytes over their own durations
        stream = {}
        for k, v in counters["kernels"].items():
            if v.get("hbm_bytes_per_launch") and v.get("counter_pass_avg_ns"):
                gbs = v["hbm_bytes_per_launch"] / v["counter_pass_avg_ns"]
                stream[k] = {"avg_us": round(v["counter_pass_avg_ns"] / 1e3, 1), "launches_per_step": v.get("kernel_trace_calls"),
                             "hbm_gbs": round(gbs, 1), "frac_of_8000": round(gbs / HBM_PEAK_GBS, 4)}
    print(json.dumps({
        "metric": "canvas Mpixels/s through Scene.render (host walk + per-node launches; result resident in HBM)",
        "value": round(h * w / dt / 1e6, 1), "unit": "Mpixels/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(dt * 1e3, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "first_render_ms": round(first_ms, 3), "cold_ms": round(min(cold), 3), "cold_with_cache_listing_ms": round(min(cold_listing), 3),
        "warm_what": "ms_per_step re-renders an unchanged document: Scene.render retains the leaf analysis and the built + planned batches "
                     "of a (scene, transform, viewport) between renders; cold_ms is a render with the cache off -- the default: nothing kept -- (best of 3), "
                     "cold_with_cache_listing_ms one with the cache on and its state dropped first (it lists the paint arrays for the guard), "
                     "first_render_ms is the process's very first render (library and kernel code loaded on the way)",
        "parity": parity,
        "host_options": {"SVGR_PAUSE_GC": os.environ.get("SVGR_PAUSE_GC"), "render_cache": "set_render_cache(4): OPT-IN, off by default",
                         "SVGR_RENDER_CACHE_TRUST": os.environ.get("SVGR_RENDER_CACHE_TRUST"),
                         "note": "a warm render compares the bytes of every paint array of the document before it reuses the retained "
                                 "batches; cold_ms is what Scene.render costs with the cache off (the default)"},
        "dtype": "f64 arithmetic and f64 layers", "data": "real asset (scene dump)",
        "config": {"workload": desc, "canvas": [h, w]},
        "roofline": {"bound": "host", "note": "no kernel binds this configuration: the step is the Python tree walk plus hundreds of "
                     "latency-bound launches; per-kernel HBM rates of the streaming kernels below", "kernels": stream,
                     "counters": {"file": counters_file, "collected_at_commit": counters.get("head"), "measured_in_this_run": False,
                                  "stale": counters.get("source_sha256") != source_sha256()}
                     if counters is not None else None},
    }))


def baseline_configs(ctx, steps=5):
    """The other BASELINE.json configurations in the driver's one line: 2 tiger @2048, 3 material-design @4096,
    5 icons.svg @4096 through `Scene.render` (the drop-in boundary: host walk + every launch, result resident in HBM), and 4's
    drawing (10 000 paths @8192) on this one GPU.  Per configuration:
      ms_per_step   the DEFAULT render: nothing kept between renders (the retained-render cache is opt-in), mean of `steps`
      warm_ms       the same render with the opt-in cache on (`set_render_cache`: leaf analysis and built + planned batches kept)
      device_ms     DEVICE time of one warm render: its launches queued up behind a hold of the stream and run back to back
                    (svgr_measure_begin / _end: HIP events on the library's stream; an upper bound if the render waits for the
                    device somewhere in the middle)
      launches      kernel launches of one default render / of one warm render (the library's own count)
      parity        the picture against pins of the REFERENCE's own render at this size (tests/golden, sparse: 16-64 k values of
                    the full canvas; float32 contract) -- config 4: rows of the canvas against the CPU oracle
    The reference prints one wall-clock figure per render (S:3854-3864): BASELINE.md holds those."""
    import numpy as np

    import svgrasterize_amd as S
    from svgrasterize_amd import _abi, scenedump, synth

    out = []
    tr = S.Transform().matrix(0, 1, 0, 1, 0, 0)

    def contract(got, ref32):
        ref32 = np.asarray(ref32, dtype=np.float32)
        a = np.abs(ref32)
        ulp = np.maximum((np.nextafter(a, np.float32(np.inf)) - a).astype(np.float64), 2.0 ** -24)
        err = np.abs(np.asarray(got, dtype=np.float64) - ref32.astype(np.float64))
        return {"values": int(err.size), "bad": int((err > ulp).sum()), "max_err": float(err.max(initial=0.0))}

    scenes = [("tiger2048", "scene_tiger.npz", "config 2: Ghostscript tiger @2048x2048 (182 solid fills incl. pre-stroked outlines)"),
              ("material4096", "scene_material.npz", "config 3: demo/material-design.svg @4096x4096 (989 fills, 935 clips)"),
              ("icons4096", "scene_icons4096.npz", "config 5: demo/icons.svg @4096x1051 (431 gradient fills, 36 blurs, 123 opacity groups)")]
    for name, fname, desc in scenes:
        rec = {"config": name, "workload": desc}
        try:
            scene, info, pins = scenedump.load_scene(os.path.join(ROOT, "tests", "golden", fname))
            h, w = info["full"]["size"]
            rec["canvas"] = [int(h), int(w)]

            def step():
                layer, _hull = scene.render(tr, viewport=[0, 0, h, w], linear_rgb=False)
                layer._device()
                return layer

            S.set_render_cache(0)
            S.clear_render_cache()
            step()
            ctx.sync()
            ts = []
            for _ in range(steps):
                ctx.sync()
                t0 = time.perf_counter()
                step()
                ctx.sync()
                ts.append((time.perf_counter() - t0) * 1e3)
            n0 = ctx.launches()
            step()
            ctx.sync()
            launches = ctx.launches() - n0
            S.set_render_cache(4)
            step()
            ctx.sync()
            tw = []
            for _ in range(steps):
                ctx.sync()
                t0 = time.perf_counter()
                step()
                ctx.sync()
                tw.append((time.perf_counter() - t0) * 1e3)
            n0 = ctx.launches()
            ctx.measure_begin(max(4.0 * (sum(tw) / len(tw)), 3.0))
            layer = step()
            dev_ms = ctx.measure_end()
            warm_launches = ctx.launches() - n0 - 1   # (the hold is a launch)
            rec.update({"ms_per_step": round(sum(ts) / len(ts), 3), "best_ms": round(min(ts), 3), "warm_ms": round(sum(tw) / len(tw), 3),
                        "device_ms": round(dev_ms, 3), "launches": int(launches), "warm_launches": int(warm_launches),
                        "canvas_mpixels_per_s": round(h * w / (sum(ts) / len(ts)) / 1e3, 1)})
            got = layer.to_canvas_f32(h, w).reshape(-1, 4)[pins["full_idx"]]
            par = contract(got, pins["full_val"])
            par["what"] = ("pins of the REFERENCE's own render of this document at this size (sha256 of its float32 canvas "
                           f"{str(info['full'].get('sha256_f32', ''))[:16]})")
            rec["parity"] = par
        except Exception as exc:  # noqa: BLE001
            rec["error"] = repr(exc)
        finally:
            S.set_render_cache(0)
            S.clear_render_cache()
        out.append(rec)
    # config 4's drawing on ONE GPU (its 8-GPU sharding is `bench.py --gpus 8`)
    rec = {"config": "synth8192", "workload": "config 4: synthetic 10 000 random closed cubic paths @ 8192x8192 on ONE GPU (sharded over N: bench.py --gpus N)"}
    try:
        from oracle import oracle as orc

        sc = synth.make_scene(8192, 10000)
        rows, cols = int(sc["viewport"][2]), int(sc["viewport"][3])
        rec["canvas"] = [rows, cols]
        b = _abi.Batch(ctx, sc["segs"], sc["seg_kind"], sc["path_seg_off"], sc["path_m6"], sc["path_rule"], sc["path_paint"], viewport=sc["viewport"])
        outb = ctx.alloc(rows * cols * 16)
        flags = _abi.RENDER_CLIP01
        ctx.sync()
        t0 = time.perf_counter()
        b.draw(outb, _abi.OUT_CANVAS_F32, flags)
        first_ms = (time.perf_counter() - t0) * 1e3
        st = b.stats
        m6 = np.array(sc["path_m6"], dtype=np.float64, copy=True)
        tn = []
        for i in range(steps):           # frames with new geometry: the reference's mode (default: nothing replayed)
            m = m6.copy()
            m[:, 2] += 0.125 * (i + 1)
            ctx.sync()
            t0 = time.perf_counter()
            b.set_transforms(m)
            b.draw(outb, _abi.OUT_CANVAS_F32, flags)
            tn.append((time.perf_counter() - t0) * 1e3)
        b.set_transforms(m6)
        b.draw(outb, _abi.OUT_CANVAS_F32, flags)
        for _ in range(3):
            b.render(outb, _abi.OUT_CANVAS_F32, flags)
        ctx.sync()
        b.timings()
        n0 = ctx.launches()
        t0 = time.perf_counter()
        k = 4 * steps
        for i in range(k):
            b.render(outb, _abi.OUT_CANVAS_F32, flags | (_abi.RENDER_TIMED if i % 4 == 0 else 0))
        ctx.sync()
        warm = (time.perf_counter() - t0) / k * 1e3
        launches = (ctx.launches() - n0) // k
        tm = b.timings()
        rec.update({"ms_per_step": round(sum(tn) / len(tn), 4), "first_frame_ms": round(first_ms, 4), "warm_ms": round(warm, 4),
                    "device_ms": round(tm["ms_total"] / max(tm["n"], 1), 4), "launches": 6, "warm_launches": int(launches),
                    "path_pixels": int(st.path_pixels), "edges": int(st.n_edges),
                    "value_warm_mpixels_per_s": round(int(st.path_pixels) / warm / 1e3, 1),
                    "what": "ms_per_step: set_transforms + svgr_batch_draw (a frame with new geometry); warm_ms: the planned replay; device_ms: HIP events around a replay"})
        acc = {"values": 0, "bad": 0, "max_err": 0.0}
        pres = synth.presentation_segs(sc)
        got_all = outb.download((rows, cols, 4), np.float32)
        for r_lo in (0, 4032, 8064):
            got = got_all[r_lo:r_lo + 128]
            ref_s, _P, _E = orc.render_solid(pres, sc["seg_kind"], sc["path_seg_off"], sc["path_rule"], sc["path_paint"],
                                             (int(sc["viewport"][0]) + r_lo, int(sc["viewport"][1]), 128, cols), clip01=True)
            c = contract(got, ref_s.astype(np.float32))
            acc["values"] += c["values"]; acc["bad"] += c["bad"]; acc["max_err"] = max(acc["max_err"], c["max_err"])
        acc["what"] = "rows 0-127, 4032-4159, 8064-8191 of a replayed frame against the CPU oracle's render of those rows (the reference's viewport cropping, S:968-971)"
        rec["parity"] = acc
        b.destroy()
    except Exception as exc:  # noqa: BLE001
        rec["error"] = repr(exc)
    out.append(rec)
    return out


def cpu_baseline(sc, budget_paths: int | None = None, strips_only: bool = False):
    """The CPU oracle (C restatement of the reference passes, oracle/svgr_oracle.c) timed on this
    host, single thread, on the same scene (or its first `budget_paths` paths).  `strips_only`: just the canvas, rendered as
    row strips on the host's cores (the checker of a second scene; nothing timed)."""
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
    # the render cut into row strips, one per thread, on this process's share of the host cores
    threads = orc.host_threads(64)
    t0 = time.perf_counter()
    rc = L.orc_render_solid_strips(*args, threads, threads)
    dt_mt = time.perf_counter() - t0
    if rc != 0:
        raise RuntimeError(f"oracle (strips) failed: {rc}")
    if strips_only:
        return canvas, None, None
    P_mt = int(stats[0])
    # ... and whole, on one thread (this canvas is the one the GPU's is checked against); an untimed census on the side counts
    # the path-pixels that are visible at all (coverage not cut to zero, S:990): the composite's algorithmic work
    canvas.fill(0.0)
    t0 = time.perf_counter()
    rc = L.orc_render_solid(*args)
    dt = time.perf_counter() - t0
    if rc != 0:
        raise RuntimeError(f"oracle failed: {rc}")
    P1 = int(stats[0])
    ref_canvas = canvas.copy() if n == n_all else None
    L.orc_visible_count(1)
    rc = L.orc_render_solid_strips(*args, threads, threads)
    visible = int(L.orc_visible_get())
    L.orc_visible_count(0)
    if rc != 0:
        raise RuntimeError(f"oracle (census) failed: {rc}")
    stats[0] = P_mt
    return ref_canvas, visible if n == n_all else None, dict(
        value=round(P1 / dt / 1e6, 3), unit="Mpixels/s (path-pixels)", cores=1, kind="port",
        sample=f"first {n} of {n_all} paths of the same scene, full viewport, {dt:.2f} s, P={P1} "
               f"(oracle/svgr_oracle.c: pass-by-pass C restatement of the reference, float64, 1 thread of {os.cpu_count()})",
        all_cores_value=round(int(stats[0]) / dt_mt / 1e6, 3), all_cores=threads,
        all_cores_what=(f"{threads} threads = the CPUs this process may run on (os.sched_getaffinity: {len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else '?'}"
                        f"; a one-GPU box hands a job its share of the host) of the host's {os.cpu_count()} logical CPUs; capped at 64"),
        all_cores_sample=f"same render as {threads} row strips (the reference's viewport cropping) on {threads} OpenMP threads, {dt_mt:.2f} s",
    )


def source_sha256() -> str:
    """sha256 of the kernel source: what the committed counter files are checked against (`counters.stale`)."""
    import hashlib

    h = hashlib.sha256()
    for f in ("svgr_hip.hip", "svgr_core.h"):
        with open(os.path.join(ROOT, "svgrasterize.py_amd", "csrc", f), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()


def load_counters(workload: str):
    """The committed per-kernel counter averages for `workload` (profiles/collect2.sh -> profiles/pmc_json.py), or None."""
    path = os.path.join(ROOT, "profiles", f"pmc_kernels_{workload}.json")
    if not os.path.exists(path):
        return None, None
    try:
        rec = json.load(open(path))
    except Exception:  # noqa: BLE001
        return None, None
    if rec.get("workload") != workload:
        return None, None
    return rec, os.path.relpath(path, ROOT)


def load_issue_table():
    """Measured cycles per VALU wave-instruction per SIMD on this chip (profiles/valu_issue.hip -> profiles/valu_issue_mi355x.json)."""
    path = os.path.join(ROOT, "profiles", "valu_issue_mi355x.json")
    try:
        t = json.load(open(path))["cycles_per_wave_instruction_per_simd"]
        at4 = {k: float(v["4_waves_per_simd"]) for k, v in t.items()}
        return at4, os.path.relpath(path, ROOT)
    except Exception:  # noqa: BLE001
        return None, None


def roofline_block(tile_ms, geo_ms, n_timed, every, P_rank, E_rank, canvas_bytes, counters, counters_file, visible=None,
                   out_kind="f32"):
    """The dominant kernel's launch, timed live (HIP events on the library's stream), against its ALGORITHMIC floors:
    the bytes that have to cross HBM (the finished canvas once + 32 B per edge) at 8 TB/s, and the composite's double
    arithmetic (8 fma + 1 add per visible path-pixel) at the measured f64 issue rate.  frac = the larger floor / t."""
    t = tile_ms * 1e-3
    alg_bytes = BYTES_PER_PATH_PIXEL * P_rank + BYTES_PER_EDGE * E_rank
    floor_bytes = canvas_bytes + BYTES_PER_EDGE * E_rank
    t_hbm = floor_bytes / (HBM_PEAK_GBS * 1e9)
    issue, issue_file = load_issue_table()
    c64 = issue["v_fma_f64"] if issue else 4.0       # cycles per f64 wave-instruction per SIMD (4 waves per SIMD)
    t_f64 = None
    if visible:
        t_f64 = 9.0 * visible / 64.0 * c64 / (N_SIMD * CLOCK_GHZ * 1e9)
    kern = None
    if counters is not None:  # (the production instantiation: float32 canvas, no clip tile; further template arguments vary)
        kern = next((v for k, v in counters["kernels"].items() if k.startswith("k_tile_render<0, false")), None)
    valu = kern.get("SQ_INSTS_VALU") if kern else None
    traffic = kern.get("hbm_bytes_per_launch") if kern else None
    bound_f64 = t_f64 is not None and t_f64 > t_hbm
    block = {
        "kernel": f"k_tile_render<{out_kind}>",
        "avg_launch_ms": round(tile_ms, 4), "geometry_ms": round(geo_ms, 4), "launches_timed": int(n_timed),
        "timing": "HIP events on the library's stream around every %d-th launch of the timed region (rank 0)" % every,
    }
    if t > 0 and not bound_f64:
        block.update({"bound": "hbm", "achieved": round(floor_bytes / t / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                      "frac": round(t_hbm / t, 4)})
    elif t > 0:
        ginst = 9.0 * visible / 64.0 / t / 1e9
        block.update({"bound": "f64-issue", "achieved": round(ginst, 1), "peak": round(N_SIMD * CLOCK_GHZ / c64, 1),
                      "unit": "G f64 wave-instructions/s", "frac": round(t_f64 / t, 4)})
    block["traffic"] = traffic
    block["floor"] = {
        "hbm_bytes": int(floor_bytes), "hbm_ms": round(t_hbm * 1e3, 4),
        "f64_ms": round(t_f64 * 1e3, 4) if t_f64 is not None else None, "visible_path_pixels": visible,
        "f64_cycles_per_wave_instruction": c64, "issue_table": issue_file,
        "note": "hbm: the canvas written once + 32 B per edge at 8 TB/s; f64: (8 fma + 1 add) per visible path-pixel / 64 lanes x "
                "measured cycles per f64 wave-instruction / (1024 SIMDs x 2.4 GHz).  frac = max of the two / launch time",
    }
    block["hbm"] = {
        "peak": HBM_PEAK_GBS, "unit": "GB/s", "traffic": traffic,
        "achieved": round(traffic / t / 1e9, 1) if traffic and t > 0 else None,
        "frac": round(traffic / t / 1e9 / HBM_PEAK_GBS, 4) if traffic and t > 0 else None,
        "note": "measured: 2 x FETCH_SIZE + WRITE_SIZE per launch (rocprofv3 --pmc, separate passes; gfx950 counts wide streaming "
                "reads at half their bytes)",
    }
    if valu and t > 0 and issue:
        f64n = sum(kern.get(k, 0.0) for k in ("SQ_INSTS_VALU_FMA_F64", "SQ_INSTS_VALU_ADD_F64", "SQ_INSTS_VALU_MUL_F64"))
        have_mix = f64n > 0
        c32 = issue["v_add_u32"]
        cyc_avail = N_SIMD * CLOCK_GHZ * 1e9 * t
        block["issue_utilisation"] = {
            "valu_wave_instructions_per_launch": int(valu), "f64_wave_instructions_per_launch": int(f64n) if have_mix else None,
            "low": round(((f64n * c64 + (valu - f64n) * c32) if have_mix else valu * c32) / cyc_avail, 4),
            "high": round(valu * c64 / cyc_avail, 4),
            "note": "executed VALU wave-instructions x measured cycles per wave-instruction / (1024 SIMDs x 2.4 GHz x t): `high` prices "
                    "every instruction like an f64 / VOP3 / DPP one (%.2f cycles), `low` the non-f64 ones like a VOP2 add (%.2f); a "
                    "utilisation of the issue slots by what was executed, not an algorithmic fraction" % (c64, c32),
        }
    block["effective_gbs"] = round(alg_bytes / t / 1e9, 1) if t > 0 else None
    block["algorithmic_bytes_per_launch"] = int(alg_bytes)
    if counters is not None:
        sha = counters.get("source_sha256")
        block["counters"] = {
            "file": counters_file, "collected_at_commit": counters.get("head"), "measured_in_this_run": False,
            "stale": (sha != source_sha256()) if sha else True,
            "of": ("rank 0 of the same N-way sharding, rendered alone on one GPU (profiles/collect_rank.sh)"
                   if "_w" in str(counters.get("workload")) else "this workload on one GPU (profiles/collect2.sh)"),
        }
        if block["counters"]["stale"]:
            block["counters"]["note"] = "the kernel source has changed since the counters were collected: counter-derived fields describe an older build"
    else:
        block["counters"] = None
    return block


def contract_counts(got32, ref64):
    """Values of the float32 canvas `got32` outside |got - f32(ref)| <= max(1 ULP_f32(ref), 2^-24), by row blocks."""
    import numpy as np

    n = bad = 0
    max_err = 0.0
    rows = got32.shape[0]
    for r0 in range(0, rows, 256):
        g = got32[r0:r0 + 256].astype(np.float64)
        r32 = ref64[r0:r0 + 256].astype(np.float32)
        a = np.abs(r32)
        ulp = np.maximum((np.nextafter(a, np.float32(np.inf)) - a).astype(np.float64), 2.0 ** -24)
        err = np.abs(g - r32.astype(np.float64))
        n += err.size
        bad += int((err > ulp).sum())
        max_err = max(max_err, float(err.max(initial=0.0)))
    return {"values": n, "bad": bad, "max_err": max_err,
            "contract": "|got - f32(ref)| <= max(1 ULP_f32(ref), 2^-24), ref = CPU oracle (f64) of the same scene, whole canvas",
            "what": "canvas of the last timed step, downloaded after the clock stopped"}


def launch_ranks(n, argv):
    """`bench.py --gpus N` started WITHOUT a launcher (no WORLD_SIZE in the environment): this process -- which has made no GPU
    kernel launch, allocation or context (counting the devices is the one runtime call it makes) and never re-execs -- starts the N ranks itself, one fresh child per GPU with RANK / LOCAL_RANK / WORLD_SIZE /
    MASTER_ADDR / MASTER_PORT set exactly as `torch.distributed.run --nnodes=1 --nproc-per-node N` would, relays rank 0's one
    JSON line, and exits non-zero if any child does.  It refuses instead of printing a one-GPU line under `"n_gpus": 1`
    when the box does not have N devices (unless the run is the one-GPU rehearsal: SVGR_BENCH_DEVICE + SVGR_BENCH_BACKEND=gloo)."""
    import socket
    import subprocess

    if "SVGR_BENCH_DEVICE" not in os.environ:
        try:
            import torch  # (device_count() does not initialise the GPU on this image)

            have = torch.cuda.device_count()
        except Exception as exc:  # noqa: BLE001
            raise SystemExit(f"[bench] --gpus {n}: cannot count the GPUs ({exc!r})")
        if have < n:
            raise SystemExit(f"[bench] --gpus {n}: this box has {have} GPU(s); not printing a smaller run under that flag "
                             "(rehearsal on one GPU: SVGR_BENCH_BACKEND=gloo SVGR_BENCH_DEVICE=0)")
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    worker = os.environ.get("SVGR_BENCH_WORKER")   # (tests: a stub in the place of this file)
    cmd = [sys.executable, worker or os.path.abspath(__file__)] + list(argv)
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL, text=True))
    # rank 0's stdout is drained on a thread while ALL children are polled: a rank that dies during start-up must not leave the
    # others sitting in init_process_group or a barrier until torch's own timeout -- on the first non-zero exit the rest are
    # terminated and the launcher fails at once
    import threading

    chunks = []
    reader = threading.Thread(target=lambda: chunks.append(procs[0].stdout.read()), daemon=True)
    reader.start()
    codes = [None] * n
    failed = False
    while any(c is None for c in codes):
        for r, p in enumerate(procs):
            if codes[r] is None:
                codes[r] = p.poll()
        if not failed and any(c not in (None, 0) for c in codes):
            failed = True
            deadline = time.monotonic() + 10.0
            for r, p in enumerate(procs):
                if codes[r] is None:
                    p.terminate()   # (exactly the children this launcher started)
        if failed and time.monotonic() > deadline:
            for r, p in enumerate(procs):
                if codes[r] is None:
                    p.kill()
        time.sleep(0.05)
    reader.join(timeout=10.0)
    out0 = chunks[0] if chunks else ""
    lines = [ln for ln in (out0 or "").splitlines() if ln.strip()]
    if any(codes):
        sys.stderr.write(f"[bench] rank exit codes {codes}\n")
        for ln in lines:
            sys.stderr.write(ln + "\n")
        # (the code of the rank that failed by itself; a child this launcher terminated reports a negative one)
        raise SystemExit(next((c for c in codes if c and c > 0), 1))
    js = [ln for ln in lines if ln.lstrip().startswith("{")]
    if len(js) != 1:
        sys.stderr.write("\n".join(lines) + "\n")
        raise SystemExit(f"[bench] rank 0 printed {len(js)} JSON lines, expected one")
    print(js[0])
    return 0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--workload", default=None, help="default: synth4096 on one GPU, synth8192 (config 4) on several")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-configs", action="store_true", help="N = 1: skip the `configs` array (the other BASELINE configurations)")
    ap.add_argument("--no-companions", action="store_true", help="N > 1: skip the single-GPU reference and the weak-scaling run")
    ap.add_argument("--cpu-paths", type=int, default=None, help="limit the CPU baseline to the first N paths")
    ap.add_argument("--time-every", type=int, default=4,
                    help="bracket the stages of every n-th timed step with HIP events (each event drains the queue for ~8 us, "
                         "so timing every step would slow the thing being measured)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return launch_ranks(args.gpus, sys.argv[1:])
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
    if args.gpus != world and rank == 0:
        print(f"[bench] --gpus {args.gpus} but WORLD_SIZE={world}; using WORLD_SIZE", file=sys.stderr)
    if args.workload is None:
        args.workload = "synth4096" if world == 1 else "synth8192"
    if args.workload in SCENE_WORKLOADS:
        if world > 1:
            raise SystemExit("the scene workloads run on one GPU")
        if args.steps == 200:
            args.steps = 20  # (tens of milliseconds per step)
        return bench_scene(args)

    import numpy as np

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

    def measure(batch, out, steps=None, active=True):
        """W warm-up steps, barrier, exactly K timed steps, barrier -> (max-over-ranks seconds, stage timings of this rank).
        `active` False: this rank only takes part in the barriers (single-GPU reference inside a multi-rank run).
        Every rank passes the same barriers and the same all-reduce whatever happens in between: a rank whose render fails
        keeps its error until they are behind it (a rank that left early would leave the others waiting in a collective)."""
        steps = args.steps if steps is None else steps
        err = None
        if active:
            try:
                for _ in range(args.warmup):
                    batch.render(out, _abi.OUT_CANVAS_F32, flags)
            except Exception as exc:  # noqa: BLE001
                err, active = exc, False
        barrier()
        if active:
            batch.timings()  # drop
        t0 = time.perf_counter()
        if active:
            try:
                for i in range(steps):
                    batch.render(out, _abi.OUT_CANVAS_F32, flags | (_abi.RENDER_TIMED if i % every == 0 else 0))
                ctx.sync()
            except Exception as exc:  # noqa: BLE001
                err, active = exc, False
        t_local = time.perf_counter() - t0
        barrier()
        tm = batch.timings() if active else dict(n=0, ms_geometry=0.0, ms_tile=0.0, ms_total=0.0)
        t_max, n_bad = (reduce([t_local], dist.ReduceOp.MAX)[0], reduce([1.0 if err else 0.0], dist.ReduceOp.SUM)[0]) if dist is not None \
            else (t_local, 1.0 if err else 0.0)
        if err is not None:
            raise err
        if n_bad:
            raise RuntimeError(f"{int(n_bad)} rank(s) failed inside the measured region")
        return t_max, tm

    def new_batch(scene):
        return _abi.Batch(ctx, scene["segs"], scene["seg_kind"], scene["path_seg_off"], scene["path_m6"], scene["path_rule"],
                          scene["path_paint"], viewport=scene["viewport"])

    sc, desc = load_workload(args.workload)
    rows, cols = int(sc["viewport"][2]), int(sc["viewport"][3])
    n_scene_paths = int(len(sc["path_seg_off"]) - 1)
    from svgrasterize_amd import dist as sdist_

    strip = int(os.environ.get("SVGR_STRIP_BANDS", str(sdist_.default_strip_bands(rows, _abi.tile_rows(), world))))  # one strip per rank

    # ---- headline: the ONE scene; N > 1: its rows sharded over the ranks (strong scaling) -----------------------------
    batch = new_batch(sc)
    st = batch.plan()
    P = P_rank = int(st.path_pixels)
    E = E_rank = int(st.n_edges)
    if world > 1:
        batch.set_bands(rank, world, strip)
        st_r = batch.plan()  # per-rank capacities: each rank keeps only the geometry that reaches its strips
        P_rank, E_rank = int(st_r.path_pixels), int(st_r.n_edges)  # (this rank's own plan: its clipped bboxes, its edges)
    own_rows = max(batch.owned_rows(), 1)
    if world > 1:
        out_t = torch.empty((own_rows, cols, 4), dtype=torch.float32, device=f"cuda:{local_rank}")
        out = ctx.wrap(out_t.data_ptr(), out_t.numel() * 4)
    else:
        out_t = None
        out = ctx.alloc(own_rows * cols * 16)
    t_max, tm = measure(batch, out)
    config = {
        "workload": desc, "canvas": [rows, cols], "paths": n_scene_paths, "edges": E, "path_pixels": P,
        "sharding": (f"{world} ranks x strips of {strip} bands ({strip * _abi.tile_rows()} rows); no data-path collective"
                     if world > 1 else "single GPU"),
    }
    scaling = "strong" if world > 1 else "weak"  # (N = 1 is the first point of either series)
    canvas_px = rows * cols
    extras = {}
    if world > 1:
        try:  # optional assembly of the full canvas on every rank (svgrasterize.py_amd/dist.py): beside the step, not in it
            from svgrasterize_amd import dist as sdist

            barrier()
            g0 = time.perf_counter()
            for _ in range(3):
                full_t = sdist.gather_canvas(out_t if coll_dev != "cpu" else out_t.cpu(), rows, _abi.tile_rows(), strip=strip)
            torch.cuda.synchronize()
            extras["all_gather_ms"] = round((time.perf_counter() - g0) / 3 * 1e3, 4)
            del full_t
        except Exception as exc:  # noqa: BLE001
            extras["all_gather_ms"] = {"error": repr(exc)}
    # ---- what a render that is NOT a replay costs (outside the timed region): the reference's only mode is a cold
    # render (S:3854-3864 times the whole of scene.render, every Path.mask flattens from scratch, S:948-957), the headline above
    # replays a plan made during warm-up
    replan_canvas = replan_m6 = None
    if world == 1 and rank == 0:
        try:
            # (three cold frames one after the other, each batch destroyed before the next is made -- a caller drawing frame after
            #  frame: the first asks the driver for fresh device memory, which is cleared before its first use; the later ones get
            #  the blocks the library's pool took back and inherit the work arrays' capacities.  `cold_ms` is the best of the five,
            #  `cold_fresh_memory_ms` the first.
            #  svgr_batch_draw = plan + render behind ONE wait; the two-call form it replaces is timed beside it)
            colds, cb = [], None
            for _ in range(5):
                if cb is not None:
                    cb.destroy()
                ctx.sync()
                c0 = time.perf_counter()
                cb = new_batch(sc)              # host arrays -> svgr_batch_create (packing + upload)
                cb.draw(out, _abi.OUT_CANVAS_F32, flags)   # census pass, wait, everything + the tile kernel, wait
                colds.append((time.perf_counter() - c0) * 1e3)
            extras["cold_ms"] = round(min(colds), 4)
            extras["cold_fresh_memory_ms"] = round(colds[0], 4)
            extras["cold_frames_ms"] = [round(c, 4) for c in colds]
            cb.destroy()
            ctx.sync()
            c0 = time.perf_counter()
            cb = new_batch(sc)
            cb.plan()
            cb.render(out, _abi.OUT_CANVAS_F32, flags)
            ctx.sync()
            extras["cold_plan_then_render_ms"] = round((time.perf_counter() - c0) * 1e3, 4)
            m6 = np.array(sc["path_m6"], dtype=np.float64, copy=True)

            def moved(i):
                m6p = m6.copy()
                m6p[:, 2] += 0.125 * (i + 1)   # every path moved by an eighth of a pixel: new edges, new bboxes, same structure
                m6p[:, 5] += 0.0625 * (i + 1)
                return m6p

            reps, t_plan, t_two = 5, 0.0, 0.0
            for i in range(reps):           # the two-call form: set_transforms + plan (waits) + render + wait
                m6p = moved(i)
                ctx.sync()
                r0 = time.perf_counter()
                cb.set_transforms(m6p)      # (voids the plan)
                p0 = time.perf_counter()
                cb.plan()
                t_plan += time.perf_counter() - p0
                cb.render(out, _abi.OUT_CANVAS_F32, flags)
                ctx.sync()
                t_two += time.perf_counter() - r0
            reps_d, t_replan, frames = 10, 0.0, []
            for i in range(reps_d):         # the frame with new geometry: set_transforms + svgr_batch_draw (one wait inside)
                m6p = moved(reps + i)
                ctx.sync()
                r0 = time.perf_counter()
                cb.set_transforms(m6p)
                cb.draw(out, _abi.OUT_CANVAS_F32, flags)
                frames.append(time.perf_counter() - r0)
                t_replan += frames[-1]
            extras["replan_frames_ms"] = [round(f * 1e3, 4) for f in frames]
            replan_m6 = m6p
            replan_canvas = out.download((own_rows, cols, 4), np.float32)   # (the last re-planned frame: checked against the oracle below)
            extras["replan_ms"] = round(t_replan / reps_d * 1e3, 4)
            extras["value_replan"] = round(P / (t_replan / reps_d) / 1e6, 1)
            extras["replan_plan_then_render_ms"] = round(t_two / reps * 1e3, 4)
            extras["plan_ms"] = round(t_plan / reps * 1e3, 4)
            extras["replan_over_step"] = round((t_replan / reps_d) / (t_max / args.steps), 2)
            extras["cold_replan_what"] = ("a frame with NEW geometry (the reference's only mode, S:948-957), host clock, outside the timed region.  cold_ms: host "
                                          "arrays -> svgr_batch_create + svgr_batch_draw (plan + render behind one wait at its end), best of five in a row "
                                          "(cold_frames_ms; each batch destroyed before the next: its device blocks are recycled and its work arrays' CAPACITIES "
                                          "inherited, so frames 2.. plan in one pass), cold_fresh_memory_ms the first of them (fresh device memory, two-pass plan); "
                                          "replan_ms: svgr_batch_set_transforms (every path moved by a fraction of a pixel) + svgr_batch_draw, mean of 10; "
                                          "value_replan = path-pixels / replan_ms; *_plan_then_render_ms: the same with svgr_batch_plan + svgr_batch_render "
                                          "+ sync (round 5's form); plan_ms: that plan call alone")
            cb.destroy()
            # (the canvas the parity check reads is the timed scene's: render it once more)
            batch.render(out, _abi.OUT_CANVAS_F32, flags)
            ctx.sync()
        except Exception as exc:  # noqa: BLE001
            extras["cold_ms"] = {"error": repr(exc)}
    got_canvas = None
    got_strips = None
    if world > 1 and rank == 0 and not args.no_cpu_baseline:
        got_strips = out_t.cpu().numpy()   # (rank 0's strips, packed in strip order; the clock has stopped)
    if world == 1 and not args.no_cpu_baseline and not args.cpu_paths:
        got_canvas = out.download((own_rows, cols, 4), np.float32)  # (the last timed step's canvas; the clock has stopped)
    del out
    out_t = None

    # ---- companions for N > 1 ---------------------------------------------------------------------------------------------
    if world > 1 and not args.no_companions:
        # (a) the same drawing on ONE GPU (rank 0 alone, the others wait at the barriers): the speed-up's denominator
        try:
            out1, prep_err = None, None
            try:  # (what only rank 0 does must not keep it from the barriers inside measure())
                batch.set_bands(0, 1, 1)
                if rank == 0:
                    batch.plan()
                    out1 = ctx.alloc(rows * cols * 16)
            except Exception as exc:  # noqa: BLE001
                prep_err = exc
            n1 = max(args.steps // 4, 10)
            s1, _tm1 = measure(batch, out1, steps=n1, active=rank == 0 and prep_err is None)
            if prep_err is not None:
                raise prep_err
            extras["single_gpu_same_scene"] = {
                "ms_per_step": round(s1 / n1 * 1e3, 4), "value": round(P / (s1 / n1) / 1e6, 1), "unit": "Mpixels/s", "steps": n1,
                "speedup_of_the_headline_over_it": round((s1 / n1) / (t_max / args.steps), 3),
            }
            out1 = None
        except Exception as exc:  # noqa: BLE001
            extras["single_gpu_same_scene"] = {"error": repr(exc)}
        # (b) weak scaling: a drawing `world` times as tall (stacked 4096-row blocks), every rank renders its own block
        try:
            wb = wout = prep_err = None
            wP_local = kept_local = 0
            try:
                wsc, wdesc = load_workload("synth4096")
                wrows, wcols = int(wsc["viewport"][2]), int(wsc["viewport"][3])
                wn = int(len(wsc["path_seg_off"]) - 1)
                tall = synth.make_tall_scene(wrows, wn, world)
                mine, kept = synth.rows_subscene(tall, rank * wrows, (rank + 1) * wrows)
                wb = new_batch(mine)
                wst = wb.plan()
                wout = ctx.alloc(wrows * wcols * 16)
                wP_local, kept_local = int(wst.path_pixels), len(kept)
            except Exception as exc:  # noqa: BLE001
                prep_err = exc
            try:
                w_max, _wtm = measure(wb, wout, active=prep_err is None)
            finally:
                wP, winst = (int(v) for v in reduce([wP_local, kept_local], dist.ReduceOp.SUM))
            if prep_err is not None:
                raise prep_err
            extras["weak_scaling"] = {
                "scaling": "weak", "value": round(wP / (w_max / args.steps) / 1e6, 1), "unit": "Mpixels/s",
                "ms_per_step": round(w_max / args.steps * 1e3, 4), "path_pixels": wP,
                "workload": f"{world} stacked blocks of: {wdesc}; each rank renders its own {wrows}-row block from the paths whose "
                            f"control points reach it ({winst} path instances in all); no data-path collective",
            }
            wout = None
            wb.destroy()
        except Exception as exc:  # noqa: BLE001
            extras["weak_scaling"] = {"error": repr(exc)}
    batch.destroy()

    if rank == 0:
        n_timed = max(tm["n"], 1)
        tile_ms = tm["ms_tile"] / n_timed
        geo_ms = tm["ms_geometry"] / n_timed
        # (N > 1: the counters of rank 0 of an N-way sharding of the same drawing, collected on one GPU by profiles/collect_rank.sh)
        counters, counters_file = load_counters(args.workload if world == 1 else f"{args.workload}_w{world}")
        line = {
            "metric": "Mpixels/sec AA coverage+composite (path-pixels/s; whole step: flatten+binning+coverage+composite; planned replay -- a frame with new geometry: value_replan)",
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
        }
        visible = None
        if world == 1 and not args.no_cpu_baseline:
            ref_canvas, visible, line["cpu_baseline"] = cpu_baseline(sc, args.cpu_paths)
            if got_canvas is not None and ref_canvas is not None:
                line["parity"] = contract_counts(got_canvas, ref_canvas)
            if replan_canvas is not None and not args.cpu_paths:
                # the last re-planned frame (every path moved, one flatten traversal, tile kernel behind the unvalidated pass) against
                # the oracle's render of the MOVED scene: whole canvas
                try:
                    ref_moved, _vis, _cb = cpu_baseline(dict(sc, path_m6=replan_m6), None, strips_only=True)
                    pr = contract_counts(replan_canvas, ref_moved)
                    pr["what"] = "canvas of the last re-planned frame (svgr_batch_set_transforms + svgr_batch_draw), against the CPU oracle's render of the moved scene"
                    line["parity_replan"] = pr
                except Exception as exc:  # noqa: BLE001
                    line["parity_replan"] = {"error": repr(exc)}
        if world > 1 and not args.no_cpu_baseline:
            # the first hardware record checks its own pixels (VERDICT r3 #4c): rank 0's strips against the oracle's render of
            # exactly those rows (the reference's own viewport cropping, S:968-971), and the CPU baseline on a bounded sample
            try:
                from oracle import oracle as orc

                pres = synth.presentation_segs(sc)
                strip_rows = strip * _abi.tile_rows()
                n_strips = (rows + strip_rows - 1) // strip_rows
                mine = [si for si in range(n_strips) if si % world == 0]
                acc = dict(values=0, bad=0, max_err=0.0)
                at = 0
                for si in mine:
                    r_lo, r_hi = si * strip_rows, min((si + 1) * strip_rows, rows)
                    ref_s, _P, _E = orc.render_solid(pres, sc["seg_kind"], sc["path_seg_off"], sc["path_rule"], sc["path_paint"],
                                                     (int(sc["viewport"][0]) + r_lo, int(sc["viewport"][1]), r_hi - r_lo, cols), clip01=True)
                    cc = contract_counts(got_strips[at:at + (r_hi - r_lo)], ref_s)
                    at += strip_rows
                    acc["values"] += cc["values"]; acc["bad"] += cc["bad"]; acc["max_err"] = max(acc["max_err"], cc["max_err"])
                acc["what"] = (f"rank 0's {len(mine)} strips of the last timed step (rows of strip s with s % {world} == 0), downloaded after the "
                               "clock stopped, against the CPU oracle's render of those rows")
                acc["contract"] = "|got - f32(ref)| <= max(1 ULP_f32(ref), 2^-24)"
                line["parity"] = acc
            except Exception as exc:  # noqa: BLE001
                line["parity"] = {"error": repr(exc)}
            try:
                n_sample = min(n_scene_paths, 2500)   # (about 10-20 s of one core on the 8192^2 scene)
                _rc, _vis, line["cpu_baseline"] = cpu_baseline(sc, args.cpu_paths or n_sample)
            except Exception as exc:  # noqa: BLE001
                line["cpu_baseline"] = {"error": repr(exc)}
        if world == 1 and args.workload == "synth4096" and not args.no_configs:
            c0 = time.perf_counter()
            try:
                line["configs"] = baseline_configs(ctx)
            except Exception as exc:  # noqa: BLE001
                line["configs"] = {"error": repr(exc)}
            line["configs_seconds"] = round(time.perf_counter() - c0, 1)
        line["roofline"] = roofline_block(tile_ms, geo_ms, tm["n"], every, P_rank, E_rank, own_rows * cols * 16, counters, counters_file,
                                          visible=visible)
        line.update(extras)
        print(json.dumps(line))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    sys.exit(main())
