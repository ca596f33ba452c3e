import cProfile, os, pstats, sys, io
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench
import svgrasterize_amd as S
from svgrasterize_amd import scenedump
wl = sys.argv[1] if len(sys.argv) > 1 else "icons4096"
fname, _ = bench.SCENE_WORKLOADS[wl]
ctx = S.Context.get(0)
scene, info, _z = scenedump.load_scene(os.path.join(ROOT, "tests", "golden", fname))
hh, ww = info["size"]
tr = S.Transform().matrix(0, 1, 0, 1, 0, 0)
for _ in range(2):
    S.clear_render_cache()
    scene.render(tr, viewport=[0, 0, hh, ww], linear_rgb=False)
ctx.sync()
pr = cProfile.Profile()
pr.enable()
for _ in range(10):
    S.clear_render_cache()
    scene.render(tr, viewport=[0, 0, hh, ww], linear_rgb=False)
ctx.sync()
pr.disable()
st = io.StringIO()
pstats.Stats(pr, stream=st).sort_stats("tottime").print_stats(32)
print(st.getvalue()[:7000])
