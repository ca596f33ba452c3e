"""cProfile of default (cache off) renders of icons.svg @4096: where the host time goes."""
import cProfile
import os
import pstats
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import svgrasterize_amd as S  # noqa: E402
from svgrasterize_amd import scenedump  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "icons4096"
scene, info, _ = scenedump.load_scene(os.path.join(ROOT, "tests", "golden", f"scene_{name}.npz"))
h, w = info["full"]["size"]
tr = S.Transform().matrix(0, 1, 0, 1, 0, 0)
ctx = S.Context.get()


def step():
    layer, _ = scene.render(tr, viewport=[0, 0, h, w], linear_rgb=False)
    layer._device()


for _ in range(3):
    step()
ctx.sync()
t0 = time.perf_counter()
for _ in range(10):
    step()
ctx.sync()
print(f"{name}: {(time.perf_counter() - t0) * 100:.3f} ms per default render")
pr = cProfile.Profile()
pr.enable()
for _ in range(10):
    step()
ctx.sync()
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(45)
