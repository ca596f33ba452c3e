#!/usr/bin/env python3
"""Draw a set of grammar-generated documents on the GPU and compare with the canvases the reference drew
(TEST INFRASTRUCTURE; the GPU-box half of the hunt).

    # build container:  SVGFUZZ_START=20000 SVGFUZZ_COUNT=200 SVGFUZZ_OUT=profiles/_tmp/hunt.npz python oracle/gen_golden.py --only svgfuzz
    # GPU box:          python oracle/hunt_generated_documents.py profiles/_tmp/hunt.npz
Full page, a random window, linear RGB and no-viewport renders, each to the float32 1-ULP contract; prints the documents
that differ (the set itself carries their text).  The committed subset lives in tests/golden/svg_fuzz_kat.npz."""
import sys, os, json, warnings
sys.path.insert(0, os.getcwd())
warnings.simplefilter("ignore")
import numpy as np
import svgrasterize_amd as S
from svgrasterize_amd import svg
z = np.load(sys.argv[1]); meta = json.loads(str(z["meta"]))
tr = S.Transform().matrix(0, 1, 0, 1, 0, 0)
nbad = 0
def cmp(got, ref, what, m):
    global nbad
    got = np.asarray(got).astype(np.float64); ref = ref.astype(np.float64)
    if got.shape != ref.shape:
        nbad += 1; print("seed", m["seed"], what, "SHAPE", got.shape, ref.shape); return
    tol = np.maximum(np.abs(np.nextafter(ref.astype(np.float32), np.float32(np.inf)).astype(np.float64) - ref), 2.0 ** -24)
    err = np.abs(got - ref)
    bad = np.argwhere((err > tol).any(axis=-1))
    if len(bad):
        nbad += 1
        print("seed", m["seed"], what, "bad px", len(bad), "max", err.max(), "rows", bad[:, 0].min(), bad[:, 0].max(), "cols", bad[:, 1].min(), bad[:, 1].max())
for k, m in enumerate(meta):
    try:
        scene, _, size = svg.svg_scene_from_str(m["text"], width=m["width"])
        h, w = m["size"]
        layer, _ = scene.render(tr, viewport=[0, 0, h, w], linear_rgb=False)
        cmp(layer.convert(pre_alpha=True, linear_rgb=False).to_canvas_f32(h, w), z[f"{k}_canvas"], "full", m)
        if m.get("crop"):
            r0, c0, rows, cols = m["crop"]
            res = scene.render(tr, viewport=[r0, c0, rows, cols], linear_rgb=False)
            win = np.zeros((rows, cols, 4), np.float32) if res is None else res[0].convert(pre_alpha=True, linear_rgb=False).translate(-r0, -c0).to_canvas_f32(rows, cols)
            cmp(win, z[f"{k}_canvas_crop"], f"crop {m['crop']}", m)
        if f"{k}_canvas_lin" in z.files:
            layer, _ = scene.render(tr, viewport=[0, 0, h, w], linear_rgb=True)
            cmp(layer.convert(pre_alpha=True, linear_rgb=True).to_canvas_f32(h, w), z[f"{k}_canvas_lin"], "linear", m)
        if m.get("free"):
            res = scene.render(tr, linear_rgb=False)
            lay = res[0].convert(pre_alpha=True, linear_rgb=False)
            if [int(lay.offset[0]), int(lay.offset[1])] != m["free"]:
                nbad += 1; print("seed", m["seed"], "free OFFSET", lay.offset, m["free"])
            else:
                cmp(lay.image.astype(np.float32), z[f"{k}_layer_free"], "free", m)
    except Exception as e:
        if "beyond +-1e9 pixels" in repr(e):  # a documented limit (32-bit pixel indices), not a mismatch
            print("seed", m["seed"], "skipped: extent beyond 1e9 pixels")
            continue
        nbad += 1
        print("seed", m["seed"], "EXCEPTION", repr(e)[:300])
print(len(meta), "documents,", nbad, "bad")
