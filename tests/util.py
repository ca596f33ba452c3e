"""Shared helpers for the parity tests."""
import json
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")


def load(name):
    return np.load(os.path.join(GOLDEN, name), allow_pickle=False)


def meta(npz, key="meta"):
    return json.loads(str(npz[key]))


def ulp_f32(ref64):
    """Size of one float32 ULP at float32(ref)."""
    r = np.abs(np.asarray(ref64, dtype=np.float32))
    return (np.nextafter(r, np.float32(np.inf)) - r).astype(np.float64)


def assert_f32_1ulp(got, ref64, what=""):
    """The north-star contract: float32 result within 1 ULP of float32(reference f64),
    with an absolute floor of 2**-24 (ULPs are meaningless next to the 1e-6 -> 0 cut)."""
    got = np.asarray(got, dtype=np.float64)
    ref32 = np.asarray(ref64, dtype=np.float32).astype(np.float64)
    assert got.shape == ref32.shape, f"{what}: shape {got.shape} != {ref32.shape}"
    tol = np.maximum(ulp_f32(ref64), 2.0 ** -24)
    err = np.abs(got - ref32)
    bad = err > tol
    assert not bad.any(), f"{what}: {int(bad.sum())} of {bad.size} values beyond 1 ULP(f32); max abs err {err.max():.3e}"


def assert_f32_1ulp_rows(got, ref64, what="", block=256):
    """assert_f32_1ulp for canvases of hundreds of megapixels: by blocks of rows (the temporaries of the whole-array form would
    be several times the canvas)."""
    assert got.shape == ref64.shape, f"{what}: shape {got.shape} != {ref64.shape}"
    n_bad, worst = 0, 0.0
    for r0 in range(0, got.shape[0], block):
        g = np.asarray(got[r0:r0 + block], dtype=np.float64)
        ref32 = np.asarray(ref64[r0:r0 + block], dtype=np.float32).astype(np.float64)
        tol = np.maximum(ulp_f32(ref64[r0:r0 + block]), 2.0 ** -24)
        err = np.abs(g - ref32)
        n_bad += int((err > tol).sum())
        worst = max(worst, float(err.max(initial=0.0)))
    assert n_bad == 0, f"{what}: {n_bad} of {got.size} values beyond 1 ULP(f32); max abs err {worst:.3e}"


def assert_close64(got, ref, atol=1e-12, what=""):
    got = np.asarray(got, dtype=np.float64)
    ref = np.asarray(ref, dtype=np.float64)
    assert got.shape == ref.shape, f"{what}: shape {got.shape} != {ref.shape}"
    err = np.abs(got - ref)
    assert err.max(initial=0.0) <= atol, f"{what}: max abs err {err.max():.3e} > {atol}"


def sort_edges(e):
    e = np.asarray(e, dtype=np.float64).reshape(-1, 4)
    return e[np.lexsort(e.T[::-1])]


def f32_contract_counts(got, ref64, what, log=True):
    """Values of `got` (float32 canvas) outside the float32 contract `|got - f32(ref)| <= max(1 ULP, 2^-24)`, sorted into
    the classes a blurred / gradient scene can produce, and recorded (gpurun_out/contract_counts.jsonl) so that the actual
    numbers are on file instead of a blanket allowance:
      tie    off by no more than 2 ULP: the reference's value sits on a float32 rounding boundary and the ~1e-16 of the
             reference's FFT blur (scipy picks FFT for every kernel size seen) against the direct double sum here decides
             which side it falls
      cut    one side is exactly 0 and the other below 4e-6: a coverage that straddles the reference's `mask < 1e-6 -> 0` cut
             (S:990) by the same ~1e-16
      other  anything else (must be empty)
    Returns dict(n, bad, tie, cut, other, max_err)."""
    got = np.asarray(got, dtype=np.float64)
    ref32 = np.asarray(ref64, dtype=np.float32).astype(np.float64)
    ulp = np.maximum(ulp_f32(ref64), 2.0 ** -24)
    err = np.abs(got - ref32)
    bad = err > ulp
    tie = bad & (err <= 2 * ulp)
    cut = bad & ~tie & ((got == 0.0) | (ref32 == 0.0)) & (np.maximum(np.abs(got), np.abs(ref32)) < 4e-6)
    other = bad & ~tie & ~cut
    rec = dict(what=what, n=int(bad.size), bad=int(bad.sum()), tie=int(tie.sum()), cut=int(cut.sum()), other=int(other.sum()),
               max_err=float(err.max(initial=0.0)))
    if log:
        try:
            os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
            with open(os.path.join(ROOT, "gpurun_out", "contract_counts.jsonl"), "a") as f:
                f.write(json.dumps(rec) + "\n")
        except OSError:
            pass
    return rec
