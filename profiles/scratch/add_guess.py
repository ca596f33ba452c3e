"""The two-pass plan's add-slot guess against what the pass used, over drawings of different character (SVGR_DBG_PLAN output)."""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CODE = r'''
import sys, os
sys.path.insert(0, %r)
import numpy as np
import svgrasterize_amd as S
from svgrasterize_amd import _abi, synth, scenedump
import bench
ctx = S.Context.get()
def run(name, sc):
    b = _abi.Batch(ctx, sc["segs"], sc["seg_kind"], sc["path_seg_off"], sc["path_m6"], sc["path_rule"], sc["path_paint"], viewport=sc["viewport"])
    sys.stderr.write("== %%s\n" %% name); sys.stderr.flush()
    b.plan(); b.destroy()
for size, n in ((4096, 4096), (8192, 10000), (2048, 12000), (1000, 20000), (4096, 2000), (3000, 6000), (8192, 3000)):
    run("synth %%d paths @ %%d" %% (n, size), synth.make_scene(size, n))
sc, _ = bench.load_workload("tiger2048")
run("tiger2048", sc)
# wide flat slivers and tall thin ones: the two extremes of pieces per edge row
rng = np.random.default_rng(3)
def rects(n, w, h, size):
    segs, off = [], [0]
    for _ in range(n):
        x0, y0 = rng.uniform(0, size - w), rng.uniform(0, size - h)
        x1, y1 = x0 + w * rng.uniform(0.5, 1), y0 + h * rng.uniform(0.5, 1)
        sk = rng.uniform(-0.3, 0.3) * h
        pts = [(x0, y0), (x1, y0 + sk), (x1, y1 + sk), (x0, y1)]
        for a, c in zip(pts, pts[1:] + pts[:1]):
            segs.append([a[1], a[0], c[1], c[0], 0, 0, 0, 0])
        off.append(len(segs))
    n_p = len(off) - 1
    return dict(segs=np.array(segs), seg_kind=np.zeros(len(segs), np.uint8), path_seg_off=np.array(off), path_m6=np.tile([1.0, 0, 0, 0, 1, 0], (n_p, 1)),
                path_rule=np.zeros(n_p, np.uint8), path_paint=np.tile([0.1, 0.2, 0.3, 0.5], (n_p, 1)), viewport=(0, 0, size, size))
run("2000 wide slivers (1500 x 6)", rects(2000, 1500, 6, 2048))
run("2000 tall slivers (6 x 1500)", rects(2000, 6, 1500, 2048))
run("30000 tiny boxes (5 x 5)", rects(30000, 5, 5, 2048))
''' % ROOT
r = subprocess.run([sys.executable, "-c", CODE], env=dict(os.environ, SVGR_DBG_PLAN="1", SVGR_NO_SPARE="1"), capture_output=True, text=True)
name = None
for ln in r.stderr.splitlines():
    if ln.startswith("== "):
        name = ln[3:]
    m = re.search(r"two passes: capacity bits (\d+) \| adds (\d+) of (\d+)", ln)
    if m:
        bits, used, cap = int(m.group(1)), int(m.group(2)), int(m.group(3))
        print(f"{name:36s} adds {used:10d} of {cap:10d} slots  ({cap / max(used, 1):.2f} x)  {'ok' if bits == 0 else 'FELL BACK (bits %d)' % bits}")
    elif "single pass" in ln and name:
        print(f"{name:36s} single pass")
if r.returncode:
    print(r.stderr[-2000:])
