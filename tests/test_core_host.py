"""CPU-side checks of the arithmetic the HIP kernels run per lane (csrc/svgr_core.h compiled for the
host by tests/host_harness.cpp) against the oracle and the reference fixtures, plus the ABI surface of
the built library.  No GPU needed; nothing here is a product code path."""
import ctypes as C
import os
import re
import subprocess

import numpy as np
import pytest

from oracle import oracle as orc
from tests.util import ROOT, load, sort_edges

HARNESS = os.path.join(ROOT, "tests", "_host_harness.so")


@pytest.fixture(scope="module")
def hh():
    src = os.path.join(ROOT, "tests", "host_harness.cpp")
    hdr = os.path.join(ROOT, "svgrasterize.py_amd", "csrc", "svgr_core.h")
    if not os.path.exists(HARNESS) or os.path.getmtime(HARNESS) < max(os.path.getmtime(src), os.path.getmtime(hdr)):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-ffp-contract=off", "-mfma", "-fPIC", "-shared", "-o", HARNESS, src])
    L = C.CDLL(HARNESS)
    f64p = np.ctypeslib.ndpointer(dtype=np.float64, flags="C_CONTIGUOUS")
    L.hh_xform.argtypes = [f64p, f64p, C.c_long, f64p]
    L.hh_flatness.argtypes = [f64p]
    L.hh_flatness.restype = C.c_double
    L.hh_flatten.argtypes = [f64p, C.c_double, f64p, C.c_long, C.c_int]
    L.hh_flatten.restype = C.c_long
    L.hh_trace_edge.argtypes = [f64p, C.c_long, C.c_long, f64p, C.c_int]
    L.hh_fill.argtypes = [C.c_double, C.c_int]
    L.hh_fill.restype = C.c_double
    L.hh_over.argtypes = [f64p, f64p]
    L.hh_key.argtypes = [C.c_double]
    L.hh_key.restype = C.c_uint64
    L.hh_unkey.argtypes = [C.c_uint64]
    L.hh_unkey.restype = C.c_double
    L.hh_bound_check.argtypes = [C.c_ulonglong, C.c_long, np.ctypeslib.ndpointer(dtype=np.int64, flags="C_CONTIGUOUS")]
    return L


def test_band_room_bounds_every_cell(hh):
    """k_path_build<2> places a cell's add list by an UPPER BOUND of its adds (svgr_core.h: band_room, summed over the edges that
    reach the cell) instead of counting them in a pass of its own.  The bound against the adds the reference's row arithmetic
    (edge_setup / row_step / row_record, S:2230-2303) leaves in every cell: a million random edges, no cell above its bound -- and
    cells that fill theirs exactly (the bound is the count, not a guess, for an edge inside one tile)."""
    out = np.zeros(3, dtype=np.int64)
    for seed in (1, 0x5F3759DF):
        hh.hh_bound_check(seed, 500000, out)
        assert out[0] > 500000 and out[1] == 0, out
        assert out[2] == 1000000, out


def test_transform_and_flatness_bit_exact(hh):
    g = load("flatten_kat.npz")
    for m, pts, out in zip(g["tr_m"], g["tr_in"], g["tr_out"]):
        got = np.empty_like(pts)
        hh.hh_xform(np.ascontiguousarray(m[:2].ravel()), np.ascontiguousarray(pts).reshape(-1), pts.size // 2, got.reshape(-1))
        assert np.array_equal(got, out)
    batch = g["rand_big_in"]
    got = np.array([hh.hh_flatness(np.ascontiguousarray(c).reshape(-1)) for c in batch])
    assert np.array_equal(got, g["rand_big_flatness"])


@pytest.mark.parametrize("mode", [0, 1, 2])
@pytest.mark.parametrize("name", ["rand_small", "rand_big", "tiny_curves", "degenerate", "tiger512"])
def test_flatten_forms_give_the_reference_edge_set(hh, name, mode):
    g = load("flatten_kat.npz")
    edges = []
    buf = np.empty(4 * 4096)
    for c in g[name + "_in"]:
        n = hh.hh_flatten(np.ascontiguousarray(c).reshape(-1), 0.1, buf, 4096, mode)
        assert 0 < n <= 4096
        edges.append(buf[: 4 * n].copy().reshape(-1, 4))
    assert np.array_equal(sort_edges(np.concatenate(edges)), sort_edges(g[name + "_edges"]))


@pytest.mark.parametrize("use_record", [0, 1])
def test_edge_rows_match_reference_coverage(hh, use_record):
    """The kernels square with x*x where the reference calls pow(x, 2): identical except for a last-bit
    difference in a fraction of a percent of the multi-pixel rows."""
    g = load("coverage_kat.npz")
    exact = 0
    for i in range(len(g["h"])):
        h, w = int(g["h"][i]), int(g["w"][i])
        ref = g["trace"][g["trace_off"][i]: g["trace_off"][i + 1]].reshape(h, w)
        got = np.zeros((h, w))
        hh.hh_trace_edge(got.reshape(-1), h, w, np.ascontiguousarray(g["lines"][i]).reshape(-1), use_record)
        assert np.allclose(got, ref, rtol=0, atol=4e-16), f"case {i}"
        exact += np.array_equal(got, ref)
    assert exact >= 0.97 * len(g["h"])


def test_fill_rules_and_over(hh):
    rng = np.random.default_rng(5)
    xs = np.concatenate([rng.uniform(-4, 4, 4000), [0.0, -0.0, 1.0, -1.0, 2.0, -2.0, 1e-7, 1 - 1e-7, 3.0000001, 1e-6, 9.99e-7]])
    for x in xs:
        nz = min(abs(x), 1.0)
        nz = 0.0 if nz < 1e-6 else nz
        eo = abs(np.remainder(x + 1.0, 2.0) - 1.0)
        eo = 0.0 if eo < 1e-6 else eo
        assert hh.hh_fill(float(x), 0) == nz
        assert hh.hh_fill(float(x), 1) == eo, x
    for _ in range(200):
        dst, src = rng.uniform(0, 1, 4), rng.uniform(0, 1, 4)
        want = src + dst * (1 - src[3])
        d = dst.copy()
        hh.hh_over(d, src)
        assert np.array_equal(d, want)


def test_coordinate_keys_are_order_preserving(hh):
    rng = np.random.default_rng(6)
    v = np.concatenate([rng.normal(0, 1e3, 500), [0.0, -0.0, 1e-300, -1e-300, 1e300, -1e300]])
    keys = np.array([hh.hh_key(float(x)) for x in v], dtype=np.uint64)
    order_v = np.argsort(v, kind="stable")
    assert (np.diff(keys[order_v].astype(np.float64)) >= 0).all()
    for x in v:
        assert hh.hh_unkey(hh.hh_key(float(x))) == x


# ------------------------------------------------------------------------------------------
# ABI surface: every symbol declared in include/svgr.h is exported by the built library
# ------------------------------------------------------------------------------------------
def test_abi_exports_every_declared_symbol():
    from svgrasterize_amd import _abi

    hdr = open(os.path.join(ROOT, "include", "svgr.h")).read()
    declared = set(re.findall(r"\b(svgr_[a-z0-9_]+)\s*\(", hdr))
    lib = _abi.load_library()  # loads without a GPU; raises if the library has not been built
    missing = [name for name in sorted(declared) if not hasattr(lib, name)]
    assert not missing, missing
    assert declared == set(_abi.EXPORTS), declared ^ set(_abi.EXPORTS)
    assert lib.svgr_abi_version() == 6
    assert lib.svgr_tile_rows() in (4, 8, 16, 32, 64) and lib.svgr_tile_cols() % 16 == 0


def test_no_gpu_means_loud_failure():
    from svgrasterize_amd import _abi

    lib = _abi.load_library()
    if lib.svgr_device_count() > 0:
        pytest.skip("a GPU is visible")
    with pytest.raises(_abi.SvgrError):
        _abi.Context(0)
