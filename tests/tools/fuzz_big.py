"""One-off stress: whole-canvas float32 parity of big synthetic scenes with other seeds / sizes (not part of the suite)."""
import os, sys
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np
import svgrasterize_amd as S
from svgrasterize_amd import _abi, synth
from oracle import oracle as orc
from util import assert_f32_1ulp_rows
ctx = S.Context.get()
for size, n, seed in ((4096, 4096, 1), (4096, 2000, 2), (3000, 6000, 3), (2048, 12000, 4), (8192, 3000, 5), (1000, 20000, 6)):
    sc = synth.make_scene(size, n, seed=synth.SEED + seed)
    batch = _abi.Batch(ctx, sc["segs"], sc["seg_kind"], sc["path_seg_off"], sc["path_m6"], sc["path_rule"], sc["path_paint"], viewport=sc["viewport"])
    st = batch.plan()
    out = ctx.alloc(size * size * 16)
    for _ in range(2):
        batch.render(out, _abi.OUT_CANVAS_F32, _abi.RENDER_CLIP01)
    got = out.download((size, size, 4), np.float32)
    ref, P, _ = orc.render_solid(synth.presentation_segs(sc), sc["seg_kind"], sc["path_seg_off"], sc["path_rule"], sc["path_paint"], sc["viewport"], clip01=True, strips=64, threads=orc.host_threads())
    assert st.path_pixels == P
    assert_f32_1ulp_rows(got, ref, what=f"{size} {n} {seed}")
    print("ok", size, n, seed, "P", P, "edges", st.n_edges, flush=True)
    batch.destroy(); del out
