"""Device memory taken by repeated Scene.render calls of a document, cache off and on (nothing may grow):  python profiles/scratch/memcheck.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.chdir(ROOT)
os.environ.setdefault("SVGR_PAUSE_GC", "1")
import torch
import bench
import svgrasterize_amd as S
from svgrasterize_amd import scenedump
ctx = S.Context.get(0)
for wl in ("icons4096", "material4096"):
    fname, _ = bench.SCENE_WORKLOADS[wl]
    scene, info, _z = scenedump.load_scene(os.path.join("tests", "golden", fname))
    hh, ww = info["size"]
    tr = S.Transform().matrix(0, 1, 0, 1, 0, 0)
    def free():
        ctx.sync(); return torch.cuda.mem_get_info(0)[0] / 2**20
    for mode, n in (("cold (cache off)", 60), ("warm (cache 4)", 300)):
        S.set_render_cache(4 if mode.startswith("warm") else 0)
        for _ in range(5): scene.render(tr, viewport=[0, 0, hh, ww], linear_rgb=False)
        f0 = free()
        marks = []
        for i in range(n):
            scene.render(tr, viewport=[0, 0, hh, ww], linear_rgb=False)
            if (i + 1) % (n // 3) == 0: marks.append(round(f0 - free(), 1))
        print(wl, mode, "device MiB taken after thirds of", n, "renders:", marks)
    S.set_render_cache(0)
