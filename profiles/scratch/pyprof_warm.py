"""cProfile of WARM Scene.render calls (retained cache on, as bench.py --workload runs them):  python profiles/scratch/pyprof_warm.py [icons4096]"""
import cProfile, os, pstats, sys, io, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ.setdefault("SVGR_PAUSE_GC", "1")
import bench
import svgrasterize_amd as S
from svgrasterize_amd import scenedump
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
t0 = time.perf_counter()
for _ in range(20):
    scene.render(tr, viewport=[0, 0, hh, ww], linear_rgb=False)
t1 = time.perf_counter()
ctx.sync()
t2 = time.perf_counter()
print("warm: host issue %.3f ms / render, + drain %.3f ms after 20" % ((t1 - t0) * 50, (t2 - t1) * 1e3))
pr = cProfile.Profile()
pr.enable()
for _ in range(20):
    scene.render(tr, viewport=[0, 0, hh, ww], linear_rgb=False)
ctx.sync()
pr.disable()
st = io.StringIO()
pstats.Stats(pr, stream=st).sort_stats("tottime").print_stats(40)
print(st.getvalue()[:9000])
