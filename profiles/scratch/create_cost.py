"""Where svgr_batch_create's 0.11-0.15 ms go: the Python constructor against the C call alone."""
import ctypes as C
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

import bench  # noqa: E402
import svgrasterize_amd as S  # noqa: E402
from svgrasterize_amd import _abi  # noqa: E402

sc, _ = bench.load_workload("synth4096")
ctx = S.Context.get(0)
lib = ctx.lib
segs = np.ascontiguousarray(sc["segs"], dtype=np.float64).reshape(-1, 8)
kind = np.ascontiguousarray(sc["seg_kind"], dtype=np.uint8)
off = np.ascontiguousarray(sc["path_seg_off"], dtype=np.int64)
m6 = np.ascontiguousarray(sc["path_m6"], dtype=np.float64).reshape(-1, 6)
rule = np.ascontiguousarray(sc["path_rule"], dtype=np.uint8)
paint = np.ascontiguousarray(sc["path_paint"], dtype=np.float64).reshape(-1, 4)
t_py, t_c = [], []
for i in range(60):
    ctx.sync()
    t0 = time.perf_counter()
    b = _abi.Batch(ctx, sc["segs"], sc["seg_kind"], sc["path_seg_off"], sc["path_m6"], sc["path_rule"], sc["path_paint"], viewport=sc["viewport"])
    t_py.append(time.perf_counter() - t0)
    ctx.sync()
    b.destroy()
    d = _abi.BatchDesc()
    d.segs, d.seg_kind, d.n_segs = _abi.ptr(segs), _abi.ptr(kind), len(segs)
    d.path_seg_off, d.n_paths = _abi.ptr(off), len(off) - 1
    d.path_m6, d.path_rule, d.path_paint = _abi.ptr(m6), _abi.ptr(rule), _abi.ptr(paint)
    d.viewport = _abi._i64x4(sc["viewport"])
    d.flatness = 0.1
    h = C.c_void_p()
    ctx.sync()
    t0 = time.perf_counter()
    rc = lib.svgr_batch_create(ctx.handle, C.byref(d), C.byref(h))
    t_c.append(time.perf_counter() - t0)
    assert rc == 0
    ctx.sync()
    lib.svgr_batch_destroy(h)
print("Batch(...) %.1f us (min %.1f) | svgr_batch_create alone %.1f us (min %.1f)" % (np.mean(t_py[10:]) * 1e6, np.min(t_py[10:]) * 1e6, np.mean(t_c[10:]) * 1e6, np.min(t_c[10:]) * 1e6))
