"""GPU parity of the residual layer (pad + DC + CDF 9/7 + SPIHT) against the CPU oracle: bit-exact."""
import numpy as np
import pytest

from tests import _lib as L

pytestmark = pytest.mark.gpu

SHAPES = [(32, 32), (33, 47), (64, 64), (100, 130), (181, 360), (37, 2047), (721, 1440)]


def _images(h, w):
    r = np.random.default_rng(h * 10007 + w)
    return np.stack([L.kat_image(h, w), r.random((h, w), dtype=np.float32), L.smooth_image(h, w, 3),
                     np.full((h, w), 0.25, np.float32)])


@pytest.mark.parametrize("shape", SHAPES)
def test_coefficients_bit_exact(shape):
    h, w = shape
    imgs = _images(h, w)
    with L.Context(len(imgs), h, w) as ctx:
        c, dc = ctx.spiht_coeffs(imgs)
    for f, img in enumerate(imgs):
        ref, rdc = L.orc_spiht_coeffs(img)
        assert dc[f] == rdc
        assert np.array_equal(c[f].reshape(ref.shape), ref), f"frame {f}"


@pytest.mark.parametrize("shape", SHAPES)
def test_streams_bit_exact(shape):
    h, w = shape
    imgs = _images(h, w)
    for tb in (0, 1024, 8 * (h * w // 20)):
        with L.Context(len(imgs), h, w) as ctx:
            got = ctx.spiht_encode(imgs, [tb] * len(imgs))
        for f, img in enumerate(imgs):
            ref = L.orc_spiht_encode(img, tb)
            assert len(got[f]) == len(ref), (tb, f)
            assert got[f] == ref, (tb, f)


@pytest.mark.parametrize("shape", SHAPES)
def test_decode_bit_exact(shape):
    h, w = shape
    imgs = _images(h, w)
    tb = 8 * (h * w // 10)
    streams = [L.orc_spiht_encode(img, tb) for img in imgs]
    with L.Context(len(imgs), h, w) as ctx:
        full = ctx.spiht_decode(streams)
        cuts = [max(17, len(s) // 3) for s in streams]
        part = ctx.spiht_decode([s[:c] for s, c in zip(streams, cuts)])
    for f, s in enumerate(streams):
        assert np.array_equal(full[f], L.orc_spiht_decode(s, h, w)), f
        assert np.array_equal(part[f], L.orc_spiht_decode(s[:cuts[f]], h, w)), f


@pytest.mark.parametrize("shape", SHAPES)
def test_prefix_reconstruction_matches_real_decode(shape):
    """The truncation-search probe (encoder bookkeeping) equals an actual decode of the prefix."""
    h, w = shape
    imgs = _images(h, w)
    tb = 8 * (h * w // 10)
    with L.Context(len(imgs), h, w) as ctx:
        streams = ctx.spiht_encode(imgs, [tb] * len(imgs))
        for frac in (1.0, 0.61, 0.33, 0.07):
            nbytes = [max(17, int(len(s) * frac)) for s in streams]
            got = ctx.spiht_decode_prefix(len(imgs), [8 * n for n in nbytes])
            for f, s in enumerate(streams):
                ref = L.orc_spiht_decode(s[:nbytes[f]], h, w, 8 * nbytes[f])
                assert np.array_equal(got[f], ref), (frac, f)


def test_dense_planes_bit_exact():
    """Every child of every set of a sweep significant on one bit-plane: 9 bits per list entry, the most a sweep of the
    encoder can produce (its packed scan fields once overflowed there)."""
    h, w = 128, 256
    yy, xx = np.mgrid[0:h, 0:w]
    imgs = np.stack([((xx + yy) % 2).astype(np.float32), ((xx // 2 + yy // 2) % 2).astype(np.float32),
                     (0.5 + 0.5 * ((xx % 2) * 2 - 1) * ((yy % 2) * 2 - 1) * (0.25 + 0.75 * ((xx * 7 + yy * 13) % 5) / 4)).astype(np.float32)])
    for tb in (0, 8 * (h * w // 2)):
        with L.Context(len(imgs), h, w) as ctx:
            got = ctx.spiht_encode(imgs, [tb] * len(imgs))
        for f, img in enumerate(imgs):
            ref = L.orc_spiht_encode(img, tb)
            assert got[f] == ref, (tb, f)
