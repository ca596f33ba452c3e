"""Library calls (C ABI entry points) of ONE warm Scene.render, by name:  python profiles/scratch/lib_calls.py [icons4096]"""
import collections, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ.setdefault("SVGR_PAUSE_GC", "1")
import bench
import svgrasterize_amd as S
from svgrasterize_amd import scenedump, _abi
wl = sys.argv[1] if len(sys.argv) > 1 else "icons4096"
fname, _ = bench.SCENE_WORKLOADS[wl]
ctx = S.Context.get(0)
S.set_render_cache(4)
scene, info, _z = scenedump.load_scene(os.path.join(ROOT, "tests", "golden", fname))
hh, ww = info["size"]
tr = S.Transform().matrix(0, 1, 0, 1, 0, 0)
for _ in range(5):
    scene.render(tr, viewport=[0, 0, hh, ww], linear_rgb=False)
ctx.sync()
counts, spent = collections.Counter(), collections.Counter()


class Proxy:
    def __init__(self, lib):
        object.__setattr__(self, "_lib", lib)

    def __getattr__(self, name):
        fn = getattr(self._lib, name)

        def call(*a):
            t = time.perf_counter()
            r = fn(*a)
            spent[name] += time.perf_counter() - t
            counts[name] += 1
            return r
        return call


real = ctx.lib
ctx.lib = Proxy(real)
N = 10
t0 = time.perf_counter()
for _ in range(N):
    scene.render(tr, viewport=[0, 0, hh, ww], linear_rgb=False)
t1 = time.perf_counter()
ctx.lib = real
ctx.sync()
print("%s: %.3f ms per render with the proxy; library calls per render %.1f, inside the library %.3f ms" % (
    wl, (t1 - t0) / N * 1e3, sum(counts.values()) / N, sum(spent.values()) / N * 1e3))
for name, c in counts.most_common():
    print("  %-34s %6.1f calls  %7.1f us each  %7.3f ms" % (name, c / N, spent[name] / c * 1e6, spent[name] / N * 1e3))
