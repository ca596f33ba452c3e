"""Wall-clock phases of a COLD Scene.render (cache off, the default):  python profiles/scratch/cold_phases.py [icons4096]"""
import collections, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ.setdefault("SVGR_PAUSE_GC", "1")
import bench
import svgrasterize_amd as S
from svgrasterize_amd import scenedump, scene as sm, geometry as gm, _abi
wl = sys.argv[1] if len(sys.argv) > 1 else "icons4096"
fname, _ = bench.SCENE_WORKLOADS[wl]
ctx = S.Context.get(0)
scene, info, _z = scenedump.load_scene(os.path.join(ROOT, "tests", "golden", fname))
hh, ww = info["size"]
tr = S.Transform().matrix(0, 1, 0, 1, 0, 0)
spent = collections.OrderedDict()


def timed(mod, name, label=None):
    fn = getattr(mod, name)
    label = label or name

    def wrapper(*a, **k):
        t = time.perf_counter()
        try:
            return fn(*a, **k)
        finally:
            spent[label] = spent.get(label, 0.0) + time.perf_counter() - t
    setattr(mod, name, wrapper)


timed(sm, "_collect_mask_jobs")
timed(sm, "_merge_runs")
timed(sm, "_plan_runs")
timed(gm, "plan_fills")
timed(_abi.Batch, "plan_many")
timed(gm, "MaskPrefetch")
timed(sm, "_prefetch_windows")
timed(sm, "build_batch")
# (_collect_mask_jobs recurses: only the outermost call counts)
depth = [0]
inner = sm._collect_mask_jobs.__wrapped__ if hasattr(sm._collect_mask_jobs, "__wrapped__") else None
for _ in range(3):
    scene.render(tr, viewport=[0, 0, hh, ww], linear_rgb=False)
ctx.sync()
N = 5
spent.clear()
tot = 0.0
for _ in range(N):
    ctx.sync()
    t0 = time.perf_counter()
    scene.render(tr, viewport=[0, 0, hh, ww], linear_rgb=False)
    t1 = time.perf_counter()
    ctx.sync()
    tot += time.perf_counter() - t0
    spent["(issue)"] = spent.get("(issue)", 0.0) + t1 - t0
print("%s cold: %.3f ms per render (with the timers)" % (wl, tot / N * 1e3))
for k, v in spent.items():
    print("  %-24s %7.3f ms" % (k, v / N * 1e3))
print("  (nested: _plan_runs contains _merge_runs, plan_fills, plan_many; _merge_runs contains build_batch; _collect_mask_jobs is recursive: its figure counts nested calls several times)")
