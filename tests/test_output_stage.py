"""Output stage: Layer.write_png / canvas_to_png against bytes written by the reference itself
(tests/golden/png_kat.npz, oracle/gen_golden.py --only png)."""
import io
import json
import os

import numpy as np
import pytest

GOLD = os.path.join(os.path.dirname(__file__), "golden")


@pytest.fixture(scope="module")
def kat():
    z = np.load(os.path.join(GOLD, "png_kat.npz"))
    return z, json.loads(str(z["meta"]))


def test_png_container_is_byte_identical(kat):
    """Host part: the same uint8 pixels give the reference's PNG file byte for byte (IHDR, one IDAT at zlib 9, IEND)."""
    from svgrasterize_amd import canvas_to_png

    z, meta = kat
    for idx, _m in enumerate(meta):
        got = canvas_to_png(z[f"{idx}_u8"]).getvalue()
        assert got == z[f"{idx}_png"].tobytes(), f"case {idx}"
    # float input is quantised like the reference does (np.round(canvas * 255))
    u8 = z["0_u8"]
    assert canvas_to_png(u8.astype(np.float64) / 255.0).getvalue() == z["0_png"].tobytes()
    out = io.BytesIO()
    assert canvas_to_png(u8, out) is out and out.getvalue() == z["0_png"].tobytes()
    with pytest.raises(ValueError):
        canvas_to_png(np.zeros((4, 4, 1)))


@pytest.mark.gpu
def test_write_png_matches_reference(kat):
    """Device part: convert(straight alpha, sRGB) + round(x * 255) on the GPU gives the reference's uint8 pixels
    (bit-exact integers) and hence its file."""
    import svgrasterize_amd as S

    z, meta = kat
    for idx, m in enumerate(meta):
        layer = S.Layer(z[f"{idx}_image"], (3, 4), pre_alpha=m["pre_alpha"], linear_rgb=m["linear_rgb"])
        u8 = layer.to_rgba8()
        assert u8.dtype == np.uint8 and u8.shape == z[f"{idx}_u8"].shape
        assert np.array_equal(u8, z[f"{idx}_u8"]), f"case {idx}: {np.argwhere(u8 != z[f'{idx}_u8'])[:5]}"
        assert layer.write_png().getvalue() == z[f"{idx}_png"].tobytes()
    with pytest.raises(ValueError):
        S.Layer(np.zeros((3, 3, 1)), (0, 0), pre_alpha=True, linear_rgb=True).write_png()
