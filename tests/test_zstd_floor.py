"""The entropy-stage shortcut of the encoder (host_pool.hip: zstd_size_lower_bound / ebcc_hip_zstd_floor).

The reference compresses every kept SPIHT prefix at zstd level 22 and then compares the size z with the pure base-layer
alternative (/root/reference/src/ebcc_codec.c:813-817, :838); the MI355X encoder skips the compression where a lower bound of z
already decides that comparison.  These tests pin the two facts the shortcut rests on against the libzstd of the image (the
one the product and the oracle dlopen): the bound never exceeds the size the library produces - at any level, on SPIHT
streams of the fixtures and on synthetic material from incompressible to degenerate - and the library
cuts its input into 128 KB blocks and nothing finer (the bound is taken block by block).  No GPU: the function is host code of the product library."""
import ctypes
import json
import os

import numpy as np
import pytest

from tests import _lib as L


def _zstd():
    for name in ("/opt/conda/lib/libzstd.so.1", "libzstd.so.1", "libzstd.so"):
        try:
            z = ctypes.CDLL(name)
            break
        except OSError:
            continue
    else:
        pytest.skip("no libzstd in this environment")
    z.ZSTD_compressBound.restype = ctypes.c_size_t
    z.ZSTD_compressBound.argtypes = [ctypes.c_size_t]
    z.ZSTD_compress.restype = ctypes.c_size_t
    z.ZSTD_compress.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_char_p, ctypes.c_size_t, ctypes.c_int]
    z.ZSTD_versionNumber.restype = ctypes.c_uint
    return z


def _compress(z, data, level):
    cap = z.ZSTD_compressBound(len(data))
    out = ctypes.create_string_buffer(cap)
    n = z.ZSTD_compress(out, cap, data, len(data), level)
    assert n <= cap
    return out.raw[:n]


def _blocks(frame):
    """(last, type, size) of every block of a zstd frame (RFC 8878 section 3.1.1)."""
    assert frame[:4] == b"\x28\xb5\x2f\xfd"
    fhd = frame[4]
    fcs_flag, single, dict_flag = fhd >> 6, (fhd >> 5) & 1, fhd & 3
    pos = 5 + (0 if single else 1) + (0, 1, 2, 4)[dict_flag] + ((1 if single else 0), 2, 4, 8)[fcs_flag]
    out = []
    while True:
        h = int.from_bytes(frame[pos:pos + 3], "little")
        last, typ, size = h & 1, (h >> 1) & 3, h >> 3
        out.append((last, typ, size))
        pos += 3 + (1 if typ == 1 else size)
        if last:
            return out


def _lib():
    if not os.path.exists(L.PRODUCT_SO):
        pytest.skip("library not built")
    lib = ctypes.CDLL(L.PRODUCT_SO)
    lib.ebcc_hip_zstd_floor.restype = ctypes.c_size_t
    lib.ebcc_hip_zstd_floor.argtypes = [ctypes.c_char_p, ctypes.c_size_t]
    return lib


def _material():
    r = np.random.default_rng(7)
    out = []
    for n in (8, 17, 100, 1000, 4096, 16384, 16385, 40000, 65536, 100000, 131072, 131073, 200000, 400000):
        out.append(("random", r.integers(0, 256, n, dtype=np.uint8).tobytes()))
        out.append(("zeros", bytes(n)))
        out.append(("period7", (bytes(range(7)) * (n // 7 + 1))[:n]))
        out.append(("low-entropy", r.choice(np.array([0, 1, 2, 255], np.uint8), n, p=[0.7, 0.1, 0.1, 0.1]).tobytes()))
        out.append(("two symbols", r.integers(0, 2, n, dtype=np.uint8).tobytes()))
        a = r.integers(0, 256, n, dtype=np.uint8)
        a[n // 3:2 * n // 3] = 0                                                      # a zero run inside noise
        out.append(("noise + run", a.tobytes()))
        k = max(1, n // 5)
        out.append(("repeated chunk", (r.integers(0, 256, k, dtype=np.uint8).tobytes() * 6)[:n]))
        bits = (r.random(8 * n) < 0.08).astype(np.uint8)                              # sparse bits: what early SPIHT planes look like
        out.append(("sparse bits", np.packbits(bits).tobytes()))
    # SPIHT streams: the residual coder's own output on smooth and noisy fields, whole and truncated
    for seed, (h, w) in enumerate(((64, 96), (128, 160), (200, 333))):
        img = (L.era5_like(h, w, seed) - 230.0) / 60.0
        img = np.clip(img + 0.02 * np.random.default_rng(seed).standard_normal((h, w)), 0, 1).astype(np.float32)
        s = L.orc_spiht_encode(img, 8 * h * w // 4)
        out.append(("spiht", s))
        out.append(("spiht prefix", s[:len(s) // 3]))
    kat = json.load(open(os.path.join(L.GOLDEN, "spiht_kat.json")))
    for case in (kat if isinstance(kat, list) else kat.get("cases", [])):
        hx = case.get("stream_hex") or case.get("stream") if isinstance(case, dict) else None
        if isinstance(hx, str) and len(hx) >= 32:
            out.append(("golden spiht", bytes.fromhex(hx)))
    return out


def test_floor_never_exceeds_what_libzstd_writes_and_small_inputs_are_one_block():
    z, lib = _zstd(), _lib()
    if z.ZSTD_versionNumber() >= 10500:
        assert lib.ebcc_hip_zstd_floor(bytes(100), 100) == 0                        # (a library that may split blocks: no shortcut)
        pytest.skip("libzstd >= 1.5: the encoder does not use the bound")
    tight = 0
    for kind, data in _material():
        floor = lib.ebcc_hip_zstd_floor(data, len(data))
        for level in (22, 19, 3, 1):
            frame = _compress(z, data, level)
            assert floor <= len(frame), (kind, len(data), level, floor, len(frame))
            # the block structure the bound is taken over: 128 KB blocks, nothing finer (libzstd < 1.5)
            assert len(_blocks(frame)) == -(-len(data) // 131072), (kind, len(data), level)
        if kind == "random" and len(data) >= 1000:
            assert floor > 0.8 * len(data)                                          # incompressible input: the bound is close to the size
            tight += 1
        if kind in ("zeros", "period7"):
            assert floor <= 16 + 3 * (len(data) // 131072)
    assert tight >= 5
    assert lib.ebcc_hip_zstd_floor(bytes((4 << 20) + 1), (4 << 20) + 1) == 0          # above 4 MB: not applicable


def test_floor_is_a_function_of_the_bytes_alone():
    """The function keeps a table between calls: every call must leave it clean."""
    lib = _lib()
    r = np.random.default_rng(3)
    a = r.integers(0, 256, 30000, dtype=np.uint8).tobytes()
    b = (r.integers(0, 4, 30000, dtype=np.uint8)).tobytes()
    fa, fb = lib.ebcc_hip_zstd_floor(a, len(a)), lib.ebcc_hip_zstd_floor(b, len(b))
    for _ in range(3):
        assert lib.ebcc_hip_zstd_floor(b, len(b)) == fb
        assert lib.ebcc_hip_zstd_floor(a, len(a)) == fa


def test_floor_on_generated_lz_material():
    """Inputs built the way an LZ parser likes them - literals from alphabets of different sizes, copies of earlier spans at
    short and long distances (overlapping ones too), runs - at many sizes: the bound must hold on every one of them."""
    z, lib = _zstd(), _lib()
    if z.ZSTD_versionNumber() >= 10500:
        pytest.skip("libzstd >= 1.5: the encoder does not use the bound")
    r = np.random.default_rng(11)
    worst = 0.0
    for case in range(160):
        n = int(r.choice([50, 300, 2000, 9000, 30000, 70000]))
        alphabet = int(r.choice([2, 4, 16, 64, 256]))
        buf = bytearray()
        while len(buf) < n:
            op = r.random()
            if op < 0.45 or len(buf) < 4:
                buf += r.integers(0, alphabet, int(r.integers(1, 40)), dtype=np.uint8).tobytes()
            elif op < 0.85:
                dist = int(r.integers(1, min(len(buf), 40000) + 1))
                ln = int(r.integers(3, 200))
                for _ in range(ln):                                                   # (byte by byte: overlapping copies repeat)
                    buf.append(buf[-dist])
            else:
                buf += bytes([int(r.integers(0, 256))]) * int(r.integers(3, 500))
        data = bytes(buf[:n])
        floor = lib.ebcc_hip_zstd_floor(data, len(data))
        for level in (22, 5):
            size = len(_compress(z, data, level))
            assert floor <= size, (case, n, alphabet, level, floor, size)
            worst = max(worst, floor / size)
    assert worst <= 1.0
