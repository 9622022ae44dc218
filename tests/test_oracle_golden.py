"""CPU tests: the oracle (oracle/*.c) against the committed golden vectors that were produced by the
reference build (oracle/make_golden.py), and - when that build is present (dev container) - directly
against it.  Bit-exact everywhere."""
import ctypes
import hashlib
import json
import os

import numpy as np
import pytest

from tests import _lib as L


def sha(b):
    return hashlib.sha256(bytes(b)).hexdigest()


def load(name):
    return json.load(open(os.path.join(L.GOLDEN, name)))


@pytest.mark.parametrize("kat", load("spiht_kat.json"), ids=lambda k: f"{k['h']}x{k['w']}")
def test_spiht_known_answers(kat):
    img = L.kat_image(kat["h"], kat["w"])
    s = L.orc_spiht_encode(img, kat["trunc_bits"])
    assert len(s) == kat["n"]
    assert sha(s) == kat["stream_sha256"]
    if "stream_hex" in kat:
        assert s.hex() == kat["stream_hex"]
    dec = L.orc_spiht_decode(s, kat["h"], kat["w"])
    assert sha(dec.tobytes()) == kat["decoded_sha256"]


def test_survey_kat_prefixes():
    """SURVEY.md section 8(c) table (sha256 prefixes measured from the reference build)."""
    want = {(32, 32, 4096): ("6a538f1114869faa910aebb7", 528), (33, 47, 8192): ("ab343752056b65205b692fa8", 1040),
            (64, 64, 0): ("812a615043f9590d1beaad27", 4249)}
    for (h, w, tb), (pre, n) in want.items():
        s = L.orc_spiht_encode(L.kat_image(h, w), tb)
        assert len(s) == n and sha(s).startswith(pre)


def test_spiht_truncated_decode_edge_cases():
    img = L.smooth_image(40, 56, 1)
    s = L.orc_spiht_encode(img, 6000)
    for nbytes in (17, 18, 40, len(s) // 2, len(s)):
        d = L.orc_spiht_decode(s[:nbytes], 40, 56, 8 * nbytes)
        assert d.shape == (40, 56) and np.isfinite(d).all() and d.min() >= 0 and d.max() <= 1
    # monotone quality: the full stream is at least as good as a short prefix
    e_full = np.abs(L.orc_spiht_decode(s, 40, 56) - img).mean()
    e_short = np.abs(L.orc_spiht_decode(s[:40], 40, 56, 320) - img).mean()
    assert e_full <= e_short


_streams = load("codec_streams.json")
_inputs = np.load(os.path.join(L.GOLDEN, "codec_inputs.npz"))


@pytest.mark.parametrize("name", sorted(_streams), ids=str)
def test_frame_codec_streams(name, monkeypatch):
    c = _streams[name]
    if c["quantile"] is None:
        monkeypatch.delenv("EBCC_INIT_BASE_ERROR_QUANTILE", raising=False)
    else:
        monkeypatch.setenv("EBCC_INIT_BASE_ERROR_QUANTILE", c["quantile"])
    L.oracle().orc_set_j2k_backend(0)
    cfg = L.make_config((1, c["h"], c["w"]), base_cr=c["base_cr"], error=c["error"], residual_type=c["mode"])
    want = bytes.fromhex(c["stream_hex"])
    got = L.orc_encode(_inputs[c["input"]], cfg)
    assert got == want
    dec = L.orc_decode(want)
    assert sha(dec.tobytes()) == c["decoded_sha256"]


_tiled = load("tiled.json")
_tiled_inputs = np.load(os.path.join(L.GOLDEN, "tiled_inputs.npz"))


@pytest.mark.parametrize("backend", [0, 1], ids=["restated-j2k", "openjpeg"])
@pytest.mark.parametrize("name", sorted(_tiled["frames"]), ids=str)
def test_multi_frame_chunk_streams(name, backend, monkeypatch):
    """Chunks of several frames: one tile per frame (reference src/ebcc_codec.c:105-180), including frame heights that
    put every tile at its own sub-band parity (the odd_* inputs)."""
    c = _tiled["frames"][name]
    for k in ("EBCC_INIT_BASE_ERROR_QUANTILE", "EBCC_DISABLE_PURE_BASE_COMPRESSION_FALLBACK"):
        monkeypatch.delenv(k, raising=False)
    if c["quantile"]:
        monkeypatch.setenv("EBCC_INIT_BASE_ERROR_QUANTILE", c["quantile"].split("+")[0])
        if c["quantile"].endswith("+nofallback"):
            monkeypatch.setenv("EBCC_DISABLE_PURE_BASE_COMPRESSION_FALLBACK", "1")
    if backend == 1 and (not hasattr(L.oracle(), "orc_opj_version") or not L.opj_version()):
        pytest.skip("OpenJPEG backend not available on this machine")
    L.oracle().orc_set_j2k_backend(backend)
    try:
        x = _tiled_inputs[c["input"]]
        cfg = L.make_config(x.shape, base_cr=c["base_cr"], error=c["error"], residual_type=c["mode"])
        want = bytes.fromhex(c["stream_hex"])
        assert L.orc_encode(x, cfg) == want
        assert sha(L.orc_decode(want).tobytes()) == c["decoded_sha256"]
    finally:
        L.oracle().orc_set_j2k_backend(0)


def test_residual_branch_is_exercised():
    assert sum(1 for c in _streams.values() if c["coeffs_size"] > 0) >= 4


def _make_data(shape):
    idx = np.indices(shape, dtype=np.float32)
    return np.ascontiguousarray(idx[0] * 100.0 + idx[1] * 1.5 + idx[2] * 0.25, dtype=np.float32)


@pytest.mark.parametrize("name", sorted(load("ebck.json")), ids=str)
def test_ebck_containers(name):
    c = load("ebck.json")[name]
    shape, chunk = tuple(c["shape"]), tuple(c["chunk"])
    cfg = L.make_config(shape, chunk if any(chunk) else None, base_cr=2.0, error=c["error"], residual_type=c["mode"])
    L.oracle().orc_set_j2k_backend(0)
    s = L.orc_encode(_make_data(shape), cfg, "orc_" + c["fn"])
    assert len(s) == c["n"] and sha(s) == c["stream_sha256"]
    d = L.orc_decode(s, "orc_ebcc_decode_chunking")
    assert sha(d.tobytes()) == c["decoded_sha256"]
    # header fields, reference tests/test_c_api.py:182-188
    assert s[:4] == b"EBCK"
    ver, ndims = np.frombuffer(s[4:12], np.uint32)
    dims = tuple(int(v) for v in np.frombuffer(s[16:40], np.uint64))
    assert ver == 1 and ndims == 3 and dims == shape


@pytest.mark.parametrize("name", sorted(_tiled["ebck"]), ids=str)
def test_multi_frame_chunk_containers(name):
    """EBCK containers whose chunks hold several frames (reference tests/test_c_api.py:194-258 shapes plus frame heights
    that are not a multiple of 32)."""
    c = _tiled["ebck"][name]
    shape, chunk = tuple(c["shape"]), tuple(c["chunk"])
    cfg = L.make_config(shape, chunk if any(chunk) else None, base_cr=2.0, error=c["error"], residual_type=c["mode"])
    L.oracle().orc_set_j2k_backend(0)
    s = L.orc_encode(_make_data(shape), cfg, "orc_ebcc_encode_chunking")
    assert len(s) == c["n"] and sha(s) == c["stream_sha256"]
    assert sha(L.orc_decode(s, "orc_ebcc_decode_chunking").tobytes()) == c["decoded_sha256"]


@pytest.mark.parametrize("name", sorted(_tiled["big"]), ids=str)
def test_multi_frame_chunk_extremes(name):
    """Two ERA5-sized frames per chunk, the most tiles a chunk can hold (63), the tallest tiles (1023 rows): formula
    inputs, hashes from the reference build."""
    c = _tiled["big"][name]
    shape = tuple(c["shape"])
    cfg = L.make_config(shape, base_cr=c["base_cr"], error=c["error"], residual_type=c["mode"])
    L.oracle().orc_set_j2k_backend(0)
    s = L.orc_encode(L.formula_frames(*shape), cfg)
    assert len(s) == c["n"] and sha(s) == c["stream_sha256"]
    assert sha(L.orc_decode(s).tobytes()) == c["decoded_sha256"]


@pytest.mark.skipif(not os.path.exists(L.REF_SO), reason="reference build only exists in the dev container")
def test_j2k_restatement_matches_openjpeg_live():
    lib = L.oracle()
    if not hasattr(lib, "orc_opj_encode"):
        pytest.skip("oracle built without the OpenJPEG backend")

    def enc(fn, img, cr):
        out = ctypes.c_void_p()
        n = fn(img.ctypes.data, img.shape[0], img.shape[1], ctypes.c_float(cr), ctypes.byref(out))
        s = ctypes.string_at(out.value, n)
        lib.orc_free(out)
        return s

    for h, w in [(32, 32), (45, 70), (100, 130)]:
        f = L.era5_like(h, w, h + w)
        img = np.ascontiguousarray((((f - f.min()) / (f.max() - f.min())) * np.float32(65535)).astype(np.uint16))
        for cr in (1.0, 4.0, 17.0, 60.0, 900.0):
            assert enc(lib.orc_opj_encode, img, cr) == enc(lib.orc_j2k_encode, img, cr), (h, w, cr)


_j2k = json.load(open(os.path.join(L.GOLDEN, "j2k_openjpeg.json")))
_j2k_inputs = np.load(os.path.join(L.GOLDEN, "j2k_inputs.npz"))


def _j2k_input(c):
    if c["input"] == "stored":
        return np.ascontiguousarray(_j2k_inputs[c["spec"]])
    h, w, k = c["spec"]
    y, x = np.mgrid[0:h, 0:w].astype(np.int64)
    v = (x * x * 3 + y * y * 5 + x * y * (k + 1)) % 4096 * 12 + ((x // 16) * 7 + (y // 16) * 13 + k) % 97 * 160 + (x * 7 + y * 13) % 31
    return np.ascontiguousarray((v % 65536).astype(np.uint16))


@pytest.mark.parametrize("i", range(len(_j2k["cases"])), ids=lambda i: "{h}x{w}-cr{cr}".format(**_j2k["cases"][i]))
def test_j2k_restatement_against_openjpeg_fixtures(i):
    """oracle/j2k_oracle.c (the restated JPEG 2000 base layer) against codestreams and decoded samples produced by the real
    OpenJPEG 2.4.0 through the reference's call sequence (oracle/make_golden_j2k.py; /root/reference/src/ebcc_codec.c:105-180,
    :1092-1136): byte-identical codestream, identical decoded samples - on any box, with or without the library."""
    c = _j2k["cases"][i]
    img = _j2k_input(c)
    s = L.orc_j2k_encode(img, c["cr"])
    assert len(s) == c["n"] and sha(s) == c["stream_sha256"]
    if "stream_hex" in c:
        assert s == bytes.fromhex(c["stream_hex"])
    assert sha(L.orc_j2k_decode(s).tobytes()) == c["decoded_sha256"]


def test_j2k_nmsedec_tables_match_openjpeg_binary_dump():
    """Spot values read from the rodata of libopenjp2 2.4.0 (lut_nmsedec_*), kept as literals."""
    lib = L.oracle()
    lib.orc_j2k_lut.restype = ctypes.POINTER(ctypes.c_int16)
    sig = [lib.orc_j2k_lut(0)[i] for i in range(128)]
    sig0 = [lib.orc_j2k_lut(1)[i] for i in range(128)]
    ref = [lib.orc_j2k_lut(2)[i] for i in range(128)]
    ref0 = [lib.orc_j2k_lut(3)[i] for i in range(128)]
    assert sig[:20] == [0] * 20 and sig[-6:] == [28416, 28800, 29184, 29568, 29952, 30336]
    assert sig0[:20] == [0, 0, 0, 0, 0, 0, 128, 128, 128, 128, 256, 256, 256, 384, 384, 512, 512, 640, 640, 768]
    assert sig0[-6:] == [29824, 30208, 30720, 31232, 31744, 32256]
    assert ref[:4] == [6144, 6016, 5888, 5760] and ref[-6:] == [5376, 5504, 5632, 5760, 5888, 6016]
    assert ref0[:4] == [8192, 7936, 7680, 7424] and ref0[-6:] == [6784, 6912, 7168, 7424, 7680, 7936]


def test_legacy_repack_is_what_the_reference_decodes():
    """tests/_lib.legacy_repack (used by the GPU legacy-stream test) against the oracle's ebcc_decode_legacy
    restatement and, in the dev container, the reference build itself."""
    import hashlib
    import json
    streams = json.load(open(os.path.join(L.GOLDEN, "codec_streams.json")))
    names = sorted(streams)[::9]
    ref = None
    if os.path.exists(L.REF_SO):
        ref = ctypes.CDLL(L.REF_SO)
        ref.ebcc_decode.restype = ctypes.c_size_t
        ref.ebcc_decode.argtypes = [ctypes.c_void_p, ctypes.c_size_t, L.c_void_pp]
    for name in names:
        s = bytes.fromhex(streams[name]["stream_hex"])
        leg = L.legacy_repack(s)
        dec = L.orc_decode(leg)
        assert hashlib.sha256(np.asarray(dec, np.float32).tobytes()).hexdigest() == streams[name]["decoded_sha256"], name
        if ref is not None:
            b = ctypes.create_string_buffer(leg, len(leg))
            out = ctypes.c_void_p()
            n = ref.ebcc_decode(b, len(leg), ctypes.byref(out))
            got = np.frombuffer(ctypes.string_at(out.value, 4 * n), np.float32)
            assert hashlib.sha256(got.tobytes()).hexdigest() == streams[name]["decoded_sha256"], name


ENV_SWITCHES = ["EBCC_DISABLE_PURE_BASE_COMPRESSION_FALLBACK", "EBCC_DISABLE_PURE_BASE_COMPRESSION_FALLBACK_CONSISTENCY",
                "EBCC_DISABLE_MEAN_ADJUSTMENT"]


def env_switch_cases():
    """(frame, config) pairs that exercise both outcomes of the :838 selection (residual kept / pure base layer)."""
    a = L.era5_like(64, 96, 21, 1.0, 0.8)
    b = L.smooth_image(96, 64, 5) if hasattr(L, "smooth_image") else L.era5_like(96, 64, 22, 2.5, 0.2)
    return [(a, L.make_config((1, 64, 96), base_cr=25.0, error=0.02, residual_type=L.MAX_ERROR)),
            (a, L.make_config((1, 64, 96), base_cr=8.0, error=2e-3, residual_type=L.RELATIVE_ERROR)),
            (np.ascontiguousarray(b, np.float32).reshape(96, 64), L.make_config((1, 96, 64), base_cr=40.0, error=0.05, residual_type=L.MAX_ERROR))]


@pytest.mark.skipif(not os.path.exists(L.REF_SO), reason="reference build only exists in the dev container")
@pytest.mark.parametrize("switch", ENV_SWITCHES)
def test_oracle_env_switches_match_reference(switch, monkeypatch):
    """The three EBCC_DISABLE_* switches of src/ebcc_codec.c:634-649: oracle == reference build, so that the GPU
    test of the same switches (tests/test_codec_gpu.py) can use the oracle on a box without the reference."""
    ref = ctypes.CDLL(L.REF_SO)
    ref.ebcc_encode.restype = ctypes.c_size_t
    ref.ebcc_encode.argtypes = [ctypes.c_void_p, ctypes.POINTER(L.CodecConfig), L.c_void_pp]
    monkeypatch.setenv(switch, "1")
    monkeypatch.setenv("EBCC_INIT_BASE_ERROR_QUANTILE", "0.1")        # loose base layer: the switches change the streams
    for frame, cfg in env_switch_cases():
        out = ctypes.c_void_p()
        n = ref.ebcc_encode(np.ascontiguousarray(frame, np.float32).ctypes.data, ctypes.byref(cfg), ctypes.byref(out))
        assert ctypes.string_at(out.value, n) == L.orc_encode(frame, cfg), switch
