import os, sys, time
sys.path.insert(0, os.getcwd())
import svgrasterize_amd as S
from svgrasterize_amd import scenedump
name = sys.argv[1]
ctx = S.Context.get(0)
swap = S.Transform().matrix(0, 1, 0, 1, 0, 0)
scene, info, z = scenedump.load_scene(os.path.join("tests", "golden", f"scene_{name}.npz"))
h, w = info["size"]
ts = []
for i in range(30):
    ctx.sync(); t0 = time.perf_counter()
    layer, _ = scene.render(swap, viewport=[0, 0, h, w], linear_rgb=False); layer._device(); ctx.sync()
    ts.append((time.perf_counter() - t0) * 1e3)
print(name, " ".join(f"{t:.1f}" for t in ts))
