#!/usr/bin/env python3
"""Which C-ABI entry points a warm Scene.render of a scene workload calls, how often and for how long (host time).

    python profiles/abi_calls.py [icons4096|material4096]      (on the GPU box)
"""
import collections
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    wl = sys.argv[1] if len(sys.argv) > 1 else "icons4096"
    import svgrasterize_amd as S
    from svgrasterize_amd import _abi, scenedump
    fname, _desc = bench.SCENE_WORKLOADS[wl]
    ctx = S.Context.get(0)
    scene, info, _z = scenedump.load_scene(os.path.join(ROOT, "tests", "golden", fname))
    hh, ww = info["size"]
    tr = S.Transform().matrix(0, 1, 0, 1, 0, 0)
    lib = _abi.load_library()
    for _ in range(3):
        scene.render(tr, viewport=[0, 0, hh, ww], linear_rgb=False)
    ctx.sync()
    calls, spent = collections.Counter(), collections.Counter()
    names = [n for n in dir(lib) if n.startswith("svgr_")]
    orig = {}
    for n in names:
        f = getattr(lib, n)
        orig[n] = f

        def wrap(*a, _f=f, _n=n):
            t0 = time.perf_counter()
            r = _f(*a)
            spent[_n] += time.perf_counter() - t0
            calls[_n] += 1
            return r

        setattr(lib, n, wrap)
    reps = 5
    t0 = time.perf_counter()
    for _ in range(reps):
        scene.render(tr, viewport=[0, 0, hh, ww], linear_rgb=False)
    ctx.sync()
    total = (time.perf_counter() - t0) / reps
    for n, f in orig.items():
        setattr(lib, n, f)
    print(f"{wl}: {total * 1e3:.2f} ms per warm render (with the wrappers); C-ABI calls per render:")
    for n, c in calls.most_common():
        print(f"  {n:34s} {c / reps:7.1f} calls  {spent[n] / reps * 1e3:7.3f} ms")
    print(f"  total {sum(calls.values()) / reps:.0f} calls, {sum(spent.values()) / reps * 1e3:.2f} ms inside the library")


if __name__ == "__main__":
    main()
