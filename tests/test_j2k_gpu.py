"""GPU parity of the JPEG 2000 base layer against the CPU oracle (itself pinned to OpenJPEG 2.4.0):
codestreams bit-exact, decoded fields bit-exact (the contract allows 1 ULP; we get 0)."""
import numpy as np
import pytest

from tests import _lib as L

pytestmark = pytest.mark.gpu

SHAPES = [(32, 32), (33, 47), (64, 96), (100, 130), (181, 360), (721, 1440)]
RATES = [1.0, 3.0, 7.5, 30.0, 120.0, 1000.0]


def _fields(h, w):
    return np.stack([L.era5_like(h, w, h + w), L.era5_like(h, w, h * w + 1, 1.0, 0.7),
                     (L.kat_image(h, w) * 40 + 260).astype(np.float32)])


@pytest.mark.parametrize("shape", SHAPES)
def test_codestreams_bit_exact(shape):
    h, w = shape
    fields = _fields(h, w)
    with L.Context(len(fields), h, w) as ctx:
        for cr in RATES:
            got, mm = ctx.j2k_encode(fields, [cr] * len(fields))
            for f, fld in enumerate(fields):
                u16, mn, mx = L.scale_u16(fld)
                assert mm[f, 0] == mn and mm[f, 1] == mx
                ref = L.orc_j2k_encode(u16, cr)
                assert len(got[f]) == len(ref), (cr, f)
                assert got[f] == ref, (cr, f)


@pytest.mark.parametrize("shape", SHAPES)
def test_emulated_decode_equals_real_decode(shape):
    h, w = shape
    fields = _fields(h, w)
    with L.Context(len(fields), h, w) as ctx:
        for cr in (2.0, 9.0, 40.0, 300.0):
            streams, mm, d = ctx.j2k_encode(fields, [cr] * len(fields), keep_device=True)
            target = [0.05, 0.3, 1.0]
            emu, nbad, esum = ctx.j2k_emulated_decode(d, len(fields), target)
            d.free()
            for f, fld in enumerate(fields):
                ref = L.map_decoded(L.orc_j2k_decode(streams[f]), mm[f, 0], mm[f, 1])
                assert np.array_equal(emu[f], ref), (cr, f)
                err = fld - ref
                assert int(nbad[f]) == int((np.abs(err) > np.float32(target[f])).sum())
                assert abs(esum[f] - err.astype(np.float64).sum()) <= 1e-6 * max(1.0, abs(esum[f]))


@pytest.mark.parametrize("shape", SHAPES)
def test_true_decode_bit_exact(shape):
    h, w = shape
    fields = _fields(h, w)
    streams, mms = [], []
    for i, fld in enumerate(fields):
        u16, mn, mx = L.scale_u16(fld)
        streams.append(L.orc_j2k_encode(u16, [1.5, 12.0, 90.0][i]))
        mms.append((mn, mx))
    with L.Context(len(fields), h, w) as ctx:
        got = ctx.j2k_decode(streams, mms)
    for f in range(len(fields)):
        ref = L.map_decoded(L.orc_j2k_decode(streams[f]), *mms[f])
        assert np.array_equal(got[f], ref), f
