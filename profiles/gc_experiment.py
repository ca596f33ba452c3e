import gc, os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import svgrasterize_amd as S
from svgrasterize_amd import scenedump
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
ctx = S.Context.get(0)
swap = S.Transform().matrix(0, 1, 0, 1, 0, 0)
for name in ("icons", "icons4096", "material"):
    scene, info, z = scenedump.load_scene(os.path.join(ROOT, "tests", "golden", f"scene_{name}.npz"))
    h, w = info["size"]
    for mode in ("gc on", "gc off"):
        if mode == "gc off": gc.disable()
        else: gc.enable()
        ts = []
        for _ in range(8):
            ctx.sync(); t0 = time.perf_counter()
            layer, _h = scene.render(swap, viewport=[0, 0, h, w], linear_rgb=False); layer._device(); ctx.sync()
            ts.append(time.perf_counter() - t0)
        print(name, mode, "best %.2f ms  median %.2f ms" % (min(ts) * 1e3, sorted(ts)[len(ts) // 2] * 1e3), flush=True)
    gc.enable()
