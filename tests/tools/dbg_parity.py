"""Where does the GPU canvas differ from the oracle?  tests/tools/dbg_parity.py [size] [paths]"""
import sys, os
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import svgrasterize_amd as S
from oracle import oracle
from svgrasterize_amd import _abi, synth

size = int(sys.argv[1]) if len(sys.argv) > 1 else 256
n = int(sys.argv[2]) if len(sys.argv) > 2 else 48
ctx = S.Context.get(0)
sc = synth.make_scene(size, n)
batch = _abi.Batch(ctx, sc["segs"], sc["seg_kind"], sc["path_seg_off"], sc["path_m6"], sc["path_rule"], sc["path_paint"], viewport=sc["viewport"])
batch.plan()
out = ctx.alloc(size * size * 16)
for rep in range(2):
    batch.render(out, _abi.OUT_CANVAS_F32, _abi.RENDER_CLIP01)
    got = out.download((size, size, 4), np.float32).astype(np.float64)
    ref, P, E = oracle.render_solid(synth.presentation_segs(sc), sc["seg_kind"], sc["path_seg_off"], sc["path_rule"], sc["path_paint"], sc["viewport"], clip01=True)
    ref32 = ref.astype(np.float32).astype(np.float64)
    ulp = np.maximum(np.nextafter(np.abs(ref32).astype(np.float32), np.float32(np.inf)).astype(np.float64) - np.abs(ref32), 2.0 ** -24)
    bad = (np.abs(got - ref32) > ulp).any(axis=2)
    print(f"render {rep}: bad pixels {int(bad.sum())} of {bad.size}")
    TR, TC = 16, 64
    for b in range((size + TR - 1) // TR):
        row = ""
        for c in range((size + TC - 1) // TC):
            t = bad[b * TR:(b + 1) * TR, c * TC:(c + 1) * TC]
            k = int(t.sum())
            row += f"{k:5d}"
        if bad[b * TR:(b + 1) * TR].any():
            rows = np.nonzero(bad[b * TR:(b + 1) * TR].any(axis=1))[0]
            print(f"band {b:3d}: {row}   bad rows in band {rows.min()}..{rows.max()}")
    ys, xs = np.nonzero(bad)
    for y, x in list(zip(ys, xs))[:6]:
        print("  px", y, x, "got", got[y, x], "ref", ref32[y, x])
