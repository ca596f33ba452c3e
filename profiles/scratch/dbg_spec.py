import os, sys
import numpy as np
sys.path.insert(0, os.getcwd())
import svgrasterize_amd as S
from svgrasterize_amd import _abi
ctx = S.Context.get()
swap = S.Transform().matrix(0, 1, 0, 1, 0, 0)
path = S.Path.from_svg("M20,10 L60,900 L100,30 L140,880 L180,15 C300,200 260,700 200,890 L20,870 Z")
segs, kinds = path.packed()
vp = [0, 0, 960, 352]
def render(tr, n=3):
    b = _abi.Batch(ctx, segs, kinds, [0, len(segs)], tr.m6(), [0], np.array([[0.2, 0.3, 0.1, 0.5]]), viewport=vp)
    st = b.plan()
    canvas = ctx.alloc(vp[2] * vp[3] * 32)
    outs = []
    for _ in range(n):
        b.render(canvas, _abi.OUT_CANVAS_F64)
        outs.append(canvas.download((vp[2], vp[3], 4), np.float64))
    return st, outs
os.environ["SVGR_NO_SPECULATIVE_PLAN"] = "1"
st, ref = render(swap)
print("staged: edges", st.n_edges, [float(np.abs(o - ref[0]).max()) for o in ref])
del os.environ["SVGR_NO_SPECULATIVE_PLAN"]
for shrink in ("1,1,1,1", "100000,1,1,1", "1,100000,1,1", "1,1,100000,1", "1,1,1,100000", "1000,1000,1000,1000"):
    os.environ["SVGR_SPEC_SHRINK"] = shrink
    st, outs = render(swap)
    d = [float(np.abs(o - ref[0]).max()) for o in outs]
    bad = [int((np.abs(o - ref[0]) > 1e-12).sum()) for o in outs]
    print(shrink, "edges", st.n_edges, "max diff vs staged", d, "n bad", bad)
    if bad[0] or bad[1]:
        k = 0 if bad[0] else 1
        w = np.argwhere(np.abs(outs[k] - ref[0]).max(axis=2) > 1e-12)
        print("   render", k, "rows", w[:, 0].min(), w[:, 0].max(), "cols", w[:, 1].min(), w[:, 1].max())
