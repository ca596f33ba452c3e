import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench
import svgrasterize_amd as S
from svgrasterize_amd import _abi
sc, desc = bench.load_workload("tiger2048")
ctx = S.Context.get(0)
b = _abi.Batch(ctx, sc["segs"], sc["seg_kind"], sc["path_seg_off"], sc["path_m6"], sc["path_rule"], sc["path_paint"], viewport=sc["viewport"])
st = b.plan()
edges, ep = b.edges()
bb = b.bboxes()
print("segs", len(sc["segs"]), "paths", len(bb), "edges", len(edges))
cnt = np.bincount(ep, minlength=len(bb))
top = np.argsort(-cnt)[:6]
for p in top:
    e = edges[ep == p]
    rows = e[:, :, 0]
    n = len(e)
    ch = [(rows[i:i + 256].min(), rows[i:i + 256].max()) for i in range(0, n, 256)]
    spans = [hi - lo for lo, hi in ch]
    print(f"path {p}: {n} edges, bbox {bb[p].tolist()}, {len(ch)} chunks, chunk row span mean {np.mean(spans):.0f} max {np.max(spans):.0f}; rows per edge mean {np.abs(rows[:,1]-rows[:,0]).mean():.1f}")
