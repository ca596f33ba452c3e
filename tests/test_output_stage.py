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


def test_parallel_png_has_the_same_pixels():
    """threads > 1: pieces deflated side by side and stitched into one zlib stream.  Not the reference's bytes, but a
    valid PNG of the same pixels (checksums verified by the decoder); threads = 1 stays the reference's exact file."""
    import struct
    import zlib

    from svgrasterize_amd.layer import canvas_to_png

    rng = np.random.default_rng(11)
    img = rng.integers(0, 256, (700, 513, 4), dtype=np.uint8)
    img[100:600, 50:400] = [12, 200, 7, 255]  # something compressible across piece borders (pieces are >= 1 MiB)
    one = canvas_to_png(img).getvalue()
    for threads, level in ((2, 9), (3, 1), (16, 6)):
        png = canvas_to_png(img, level=level, threads=threads).getvalue()
        assert png[:8] == one[:8] and png[8:33] == one[8:33]  # signature + IHDR
        pos, idat = 8, b""
        while pos < len(png):
            n, tag = struct.unpack(">I4s", png[pos:pos + 8])
            body = png[pos + 8: pos + 8 + n]
            assert struct.unpack(">I", png[pos + 8 + n: pos + 12 + n])[0] == zlib.crc32(body, zlib.crc32(tag)) & 0xFFFFFFFF
            if tag == b"IDAT":
                idat += body
            pos += 12 + n
        raw = np.frombuffer(zlib.decompress(idat), dtype=np.uint8).reshape(700, 513 * 4 + 1)
        assert not raw[:, 0].any() and np.array_equal(raw[:, 1:].reshape(700, 513, 4), img)
    assert canvas_to_png(np.zeros((0, 5, 4), np.uint8), threads=4).getvalue()[:8] == one[:8]  # empty image: one empty piece
