#!/usr/bin/env python3
"""Generate tests/golden/* from the REFERENCE build (oracle/_ref/*.so = /root/reference sources compiled
unmodified + the image's OpenJPEG 2.4.0 / zstd 1.4.9).  Run in the dev container only:

    make -C oracle ref && python oracle/make_golden.py

Fixtures are data only: inputs (stored, or defined by an integer formula) and the bytes / hashes the
reference produced for them.  Test infrastructure; never imported by the product.
"""
import ctypes
import hashlib
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tests import _lib as L  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")
os.makedirs(OUT, exist_ok=True)

ref = ctypes.CDLL(L.REF_SO)
sp = ctypes.CDLL(L.REF_SPIHT_SO)
sp.spiht_encode.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_size_t, L.c_void_pp, L.c_size_p,
                            ctypes.c_size_t, ctypes.c_size_t]
sp.spiht_decode.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_size_t,
                            ctypes.c_size_t]
for name in ("ebcc_encode", "ebcc_encode_chunking", "ebcc_encode_chunking_compat"):
    f = getattr(ref, name)
    f.restype = ctypes.c_size_t
    f.argtypes = [ctypes.c_void_p, ctypes.POINTER(L.CodecConfig), L.c_void_pp]
for name in ("ebcc_decode", "ebcc_decode_chunking"):
    f = getattr(ref, name)
    f.restype = ctypes.c_size_t
    f.argtypes = [ctypes.c_void_p, ctypes.c_size_t, L.c_void_pp]
ref.free_buffer.argtypes = [ctypes.c_void_p]


def sha(b):
    return hashlib.sha256(bytes(b)).hexdigest()


def ref_spiht(img, tb):
    h, w = img.shape
    buf, n = ctypes.c_void_p(), ctypes.c_size_t()
    sp.spiht_encode(img.ctypes.data, h, w, ctypes.byref(buf), ctypes.byref(n), tb, 3)
    s = ctypes.string_at(buf.value, n.value)
    out = np.zeros((h, w), np.float32)
    sp.spiht_decode(buf.value, n.value, out.ctypes.data, h, w, 8 * n.value)
    return s, out


def ref_encode(data, cfg, fn="ebcc_encode"):
    data = np.ascontiguousarray(data, np.float32)
    out = ctypes.c_void_p()
    n = getattr(ref, fn)(data.ctypes.data, ctypes.byref(cfg), ctypes.byref(out))
    s = ctypes.string_at(out.value, n)
    ref.free_buffer(out)
    return s


def ref_decode(s, fn="ebcc_decode"):
    b = ctypes.create_string_buffer(bytes(s), len(s))
    out = ctypes.c_void_p()
    n = getattr(ref, fn)(b, len(s), ctypes.byref(out))
    a = np.frombuffer(ctypes.string_at(out.value, 4 * n), np.float32).copy()
    ref.free_buffer(out)
    return a


def main():
    # ---- 1. SPIHT known-answer vectors
    kats = []
    for h, w, tb in [(32, 32, 4096), (33, 47, 8192), (64, 64, 0), (721, 1440, 370096)]:
        img = L.kat_image(h, w)
        s, dec = ref_spiht(img, tb)
        e = {"h": h, "w": w, "trunc_bits": tb, "n": len(s), "stream_sha256": sha(s), "decoded_sha256": sha(dec.tobytes())}
        if h * w <= 4096:
            e["stream_hex"] = s.hex()
        kats.append(e)
    json.dump(kats, open(os.path.join(OUT, "spiht_kat.json"), "w"), indent=1)

    # ---- 2. frame codec streams for small stored inputs, every mode, with and without the residual kept
    cases = {}
    arrays = {}
    k = 0
    for (h, w, seed) in [(64, 96, 64), (100, 130, 100)]:
        field = L.era5_like(h, w, seed)
        rough = L.era5_like(h, w, seed + 1, 1.0, 0.7)
        arrays[f"in{k}"] = field
        arrays[f"in{k + 1}"] = rough
        for key, arr in ((f"in{k}", field), (f"in{k + 1}", rough)):
            for cr, mode, err in [(10, 0, 0.0), (30, 1, 0.5), (30, 1, 0.1), (30, 2, 1e-3), (5, 1, 0.01), (100, 1, 2.0)]:
                for q in (None, "0.02", "0.1"):
                    if q is None:
                        os.environ.pop("EBCC_INIT_BASE_ERROR_QUANTILE", None)
                    else:
                        os.environ["EBCC_INIT_BASE_ERROR_QUANTILE"] = q
                    cfg = L.make_config((1, h, w), base_cr=cr, error=err, residual_type=mode)
                    s = ref_encode(arr, cfg)
                    d = ref_decode(s)
                    name = f"{key}_cr{cr}_m{mode}_e{err}_q{q}"
                    cases[name] = {"input": key, "h": h, "w": w, "base_cr": cr, "mode": mode, "error": err,
                                   "quantile": q, "stream_hex": s.hex(), "decoded_sha256": sha(d.tobytes()),
                                   "coeffs_size": int(np.frombuffer(s[16:24], np.uint64)[0])}
        k += 2
    os.environ.pop("EBCC_INIT_BASE_ERROR_QUANTILE", None)
    const = np.full((64, 64), 3.25, np.float32)
    arrays["const"] = const
    cfg = L.make_config((1, 64, 64), base_cr=10, error=0.1, residual_type=1)
    s = ref_encode(const, cfg)
    cases["const"] = {"input": "const", "h": 64, "w": 64, "base_cr": 10, "mode": 1, "error": 0.1, "quantile": None,
                      "stream_hex": s.hex(), "decoded_sha256": sha(ref_decode(s).tobytes()), "coeffs_size": 0}
    np.savez_compressed(os.path.join(OUT, "codec_inputs.npz"), **arrays)
    json.dump(cases, open(os.path.join(OUT, "codec_streams.json"), "w"), indent=0)

    # ---- 3. EBCK containers (reference tests/test_c_api.py data formula, one frame per chunk)
    def make_data(shape):
        idx = np.indices(shape, dtype=np.float32)
        return np.ascontiguousarray(idx[0] * 100.0 + idx[1] * 1.5 + idx[2] * 0.25, dtype=np.float32)

    ebck = {}
    for shape, chunk, fn in [((2, 32, 32), (1, 32, 32), "ebcc_encode_chunking"),
                             ((2, 33, 35), (1, 64, 64), "ebcc_encode_chunking"),
                             ((3, 40, 50), (1, 32, 32), "ebcc_encode_chunking"),
                             ((2, 32, 32), (0, 0, 0), "ebcc_encode_chunking_compat")]:
        for mode, err in ((1, 0.01), (2, 0.01), (0, 0.0)):
            cfg = L.make_config(shape, chunk if any(chunk) else None, base_cr=2.0, error=err, residual_type=mode)
            s = ref_encode(make_data(shape), cfg, fn)
            d = ref_decode(s, "ebcc_decode_chunking")
            ebck[f"{fn}_{shape}_{chunk}_m{mode}"] = {"shape": shape, "chunk": chunk, "fn": fn, "mode": mode, "error": err,
                                                     "stream_sha256": sha(s), "n": len(s),
                                                     "decoded_sha256": sha(d.tobytes())}
    json.dump(ebck, open(os.path.join(OUT, "ebck.json"), "w"), indent=1)

    # ---- 4. full-size stream hashes on integer-formula inputs (exactly reproducible anywhere)
    big = {}
    y, x = np.mgrid[0:721, 0:1440]
    f1 = (250.0 + ((x * 3 + y * 5) % 1024).astype(np.float32) / np.float32(64.0)
          + (((x // 16) * 7 + (y // 16) * 13) % 97).astype(np.float32)).astype(np.float32)
    for cr, mode, err in [(100, 0, 0.0), (30, 1, 0.5)]:
        cfg = L.make_config((1, 721, 1440), base_cr=cr, error=err, residual_type=mode)
        s = ref_encode(f1, cfg)
        big[f"formula1_cr{cr}_m{mode}"] = {"base_cr": cr, "mode": mode, "error": err, "n": len(s), "stream_sha256": sha(s),
                                           "decoded_sha256": sha(ref_decode(s).tobytes())}
    json.dump(big, open(os.path.join(OUT, "codec_big.json"), "w"), indent=1)
    print("golden fixtures written to", OUT)


def tiled():
    """Chunks holding several frames (reference src/ebcc_codec.c:105-180: one multi-tile JPEG 2000 image per chunk)
    -> tests/golden/tiled.json + tiled_inputs.npz.  `python oracle/make_golden.py tiled`"""
    rng = np.random.default_rng(77)
    idx = lambda shape: np.indices(shape, dtype=np.float32)
    inputs = {
        "ramp_2x32x32": np.ascontiguousarray(idx((2, 32, 32))[0] * 100.0 + idx((2, 32, 32))[1] * 1.5 + idx((2, 32, 32))[2] * 0.25, np.float32),
        "noise_4x32x64": (270 + 5 * np.cumsum(rng.standard_normal((4, 32, 64)), axis=2) / 8 + rng.standard_normal((4, 32, 64))).astype(np.float32),
        "waves_2x64x96": np.stack([(10 * np.sin(np.arange(96)[None, :] / (7.0 + k)) * np.cos(np.arange(64)[:, None] / (5.0 + k))
                                    + 0.3 * rng.standard_normal((64, 96))) for k in range(2)]).astype(np.float32),
        "mixed_3x128x160": np.stack([L.era5_like(128, 160, 40 + k, 1.3, 1.5) for k in range(3)]).astype(np.float32),
        "const_2x32x32": np.full((2, 32, 32), 4.5, np.float32),
        # frame heights that are not a multiple of 32: every tile has its own sub-band extents and low/high parity
        "odd_2x33x35": np.stack([L.era5_like(33, 35, 60 + k, 1.4, 2.0) for k in range(2)]).astype(np.float32),
        "odd_3x37x70": np.stack([L.era5_like(37, 70, 63 + k, 1.2, 1.0) for k in range(3)]).astype(np.float32),
        "odd_2x100x130": np.stack([L.era5_like(100, 130, 66 + k, 1.6, 3.0) for k in range(2)]).astype(np.float32),
        "odd_5x65x129": np.stack([L.era5_like(65, 129, 70 + k, 1.3, 2.5) for k in range(5)]).astype(np.float32),
    }
    np.savez_compressed(os.path.join(OUT, "tiled_inputs.npz"), **inputs)
    cases = {}
    for name, base_cr, mode, err, quant in [("ramp_2x32x32", 2.0, 0, 0.0, None), ("ramp_2x32x32", 2.0, 1, 0.01, None),
                                            ("ramp_2x32x32", 2.0, 2, 0.01, None), ("noise_4x32x64", 10.0, 1, 0.05, None),
                                            ("noise_4x32x64", 10.0, 2, 1e-3, None), ("noise_4x32x64", 40.0, 0, 0.0, None),
                                            ("waves_2x64x96", 20.0, 1, 0.02, "0.1"), ("waves_2x64x96", 20.0, 1, 0.02, None),
                                            ("waves_2x64x96", 6.0, 2, 5e-4, "0.02"), ("mixed_3x128x160", 30.0, 1, 0.1, None),
                                            ("mixed_3x128x160", 30.0, 1, 0.1, "0.1"), ("const_2x32x32", 5.0, 1, 0.1, None),
                                            # residual layer kept (the pure-base fallback switched off): SPIHT over the whole chunk
                                            ("waves_2x64x96", 20.0, 1, 0.02, "0.1+nofallback"), ("mixed_3x128x160", 30.0, 1, 0.1, "0.1+nofallback"),
                                            ("noise_4x32x64", 10.0, 2, 1e-3, "0.02+nofallback"),
                                            ("odd_2x33x35", 2.0, 0, 0.0, None), ("odd_2x33x35", 6.0, 1, 0.02, None),
                                            ("odd_2x33x35", 6.0, 1, 0.02, "0.1+nofallback"), ("odd_3x37x70", 5.0, 1, 0.05, None),
                                            ("odd_3x37x70", 15.0, 2, 2e-3, "0.05+nofallback"), ("odd_2x100x130", 20.0, 2, 1e-3, None),
                                            ("odd_2x100x130", 40.0, 1, 0.05, "0.1+nofallback"), ("odd_2x100x130", 40.0, 0, 0.0, None),
                                            ("odd_5x65x129", 8.0, 1, 0.02, None), ("odd_5x65x129", 30.0, 1, 0.1, "0.2+nofallback")]:
        os.environ.pop("EBCC_INIT_BASE_ERROR_QUANTILE", None)
        os.environ.pop("EBCC_DISABLE_PURE_BASE_COMPRESSION_FALLBACK", None)
        if quant:
            os.environ["EBCC_INIT_BASE_ERROR_QUANTILE"] = quant.split("+")[0]
            if quant.endswith("+nofallback"):
                os.environ["EBCC_DISABLE_PURE_BASE_COMPRESSION_FALLBACK"] = "1"
        x = inputs[name]
        cfg = L.make_config(x.shape, base_cr=base_cr, error=err, residual_type=mode)
        s = ref_encode(x, cfg)
        d = ref_decode(s)
        import struct
        coeffs = struct.unpack("<Q", s[16:24])[0]
        cases[f"{name}_cr{base_cr:g}_m{mode}_q{quant}"] = {"input": name, "base_cr": base_cr, "mode": mode, "error": err, "quantile": quant,
                                                          "stream_hex": s.hex(), "decoded_sha256": sha(d.tobytes()), "coeffs_size": coeffs}
        print(name, mode, quant, len(s), "coeffs", coeffs, "max err", float(np.abs(d - x.ravel()).max()))
    os.environ.pop("EBCC_INIT_BASE_ERROR_QUANTILE", None)
    os.environ.pop("EBCC_DISABLE_PURE_BASE_COMPRESSION_FALLBACK", None)
    make_data = lambda shape: np.ascontiguousarray(idx(shape)[0] * 100.0 + idx(shape)[1] * 1.5 + idx(shape)[2] * 0.25, np.float32)
    ebck = {}
    for shape, chunk in [((3, 33, 35), (2, 32, 32)), ((2, 32, 32), (4, 32, 32)), ((2, 32, 32), (0, 0, 0)), ((5, 64, 64), (2, 64, 32)),
                         ((5, 45, 91), (2, 45, 91)), ((4, 100, 130), (3, 50, 65)), ((3, 33, 35), (0, 0, 0)), ((6, 40, 48), (6, 40, 48))]:
        for mode, err in ((1, 0.01), (0, 0.0)):
            cfg = L.make_config(shape, chunk if any(chunk) else None, base_cr=2.0, error=err, residual_type=mode)
            s = ref_encode(make_data(shape), cfg, "ebcc_encode_chunking")
            d = ref_decode(s, "ebcc_decode_chunking")
            ebck[f"{shape}_{chunk}_m{mode}"] = {"shape": shape, "chunk": chunk, "mode": mode, "error": err, "n": len(s),
                                                "stream_sha256": sha(s), "decoded_sha256": sha(d.tobytes())}
    # full-size and extreme multi-frame chunks on formula inputs (hashes only): two ERA5-sized frames per chunk, the
    # most tiles a chunk can hold (63 x 32 rows), the tallest tiles (2 x 1023 rows)
    big = {}
    for shape, cr, mode, err in [((2, 721, 1440), 30.0, 1, 0.5), ((2, 721, 1440), 100.0, 0, 0.0), ((63, 32, 40), 4.0, 1, 0.05),
                                 ((62, 33, 48), 4.0, 1, 0.05), ((2, 1023, 33), 6.0, 2, 1e-3)]:
        cfg = L.make_config(shape, base_cr=cr, error=err, residual_type=mode)
        s = ref_encode(L.formula_frames(*shape), cfg)
        big["x".join(map(str, shape)) + f"_cr{cr:g}_m{mode}"] = {"shape": shape, "base_cr": cr, "mode": mode, "error": err, "n": len(s),
                                                               "stream_sha256": sha(s), "decoded_sha256": sha(ref_decode(s).tobytes())}
        print(shape, cr, mode, len(s), flush=True)
    json.dump({"frames": cases, "ebck": ebck, "big": big}, open(os.path.join(OUT, "tiled.json"), "w"), indent=0)
    print("tiled fixtures written")


if __name__ == "__main__":
    tiled() if sys.argv[1:] == ["tiled"] else main()
