import os, sys, json, numpy as np
sys.path.insert(0, os.getcwd())
import svgrasterize_amd as S
from svgrasterize_amd import scenedump, _abi
from svgrasterize_amd.scene import build_batch
scene, info, z = scenedump.load_scene("tests/golden/scene_tiger.npz")
r = info["renders"][0]; tr = S.Transform().matrix(0,1,0,1,0,0).scale(r["scale"]); hh, ww = r["size"]
img, st = S.render_canvas(scene, tr, [0,0,hh,ww], linear_rgb=False, out_f64=True, clip01=False)
ref = z["s128_canvas"]
err = np.abs(img-ref).max(axis=2)
ys, xs = np.nonzero(err > 1e-9)
print("n bad", len(ys), "rows", sorted(set(ys))[:20], "cols", sorted(set(xs))[:40])
for y,x in list(zip(ys,xs))[:10]:
    print(y,x,img[y,x],ref[y,x])
leaves = scene.leaves(tr, False)
b = build_batch(leaves,[0,0,hh,ww]); b.plan(); bb=b.bboxes()
print("bboxes containing first bad px:")
if len(ys):
    y,x=ys[0],xs[0]
    for i,(r0,c0,rr,cc) in enumerate(bb):
        if rr>0 and r0<=y<r0+rr: print(i,(r0,c0,rr,cc), 'covers col' if c0<=x<c0+cc else 'row only')
