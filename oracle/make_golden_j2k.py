#!/usr/bin/env python3
"""Golden vectors for the JPEG 2000 base layer ALONE, generated with the real OpenJPEG 2.4.0 of this image through the
oracle's optional backend (oracle/opj_backend.c: the reference's own call sequence, /root/reference/src/ebcc_codec.c:105-180
encode and :1092-1136 decode, on a u16 image).  TEST INFRASTRUCTURE; run in the dev container only:

    python3 oracle/make_golden_j2k.py        ->  tests/golden/j2k_openjpeg.json, tests/golden/j2k_inputs.npz

tests/test_oracle_golden.py::test_j2k_restatement_against_openjpeg_fixtures pins oracle/j2k_oracle.c on them on any box
(the live comparison with the library only runs where the library is).  Inputs are integer formulas (reproducible
anywhere) or stored arrays; small codestreams are stored whole, large ones as sha256."""
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


def formula_u16(h, w, k=0):
    """smooth ramps + blocks + fine texture, exactly reproducible"""
    y, x = np.mgrid[0:h, 0:w].astype(np.int64)
    v = (x * x * 3 + y * y * 5 + x * y * (k + 1)) % 4096 * 12 + ((x // 16) * 7 + (y // 16) * 13 + k) % 97 * 160 + (x * 7 + y * 13) % 31
    return np.ascontiguousarray((v % 65536).astype(np.uint16))


def main():
    lib = L.oracle()
    assert hasattr(lib, "orc_opj_encode") and L.opj_version().startswith("2.4.0"), "needs the OpenJPEG 2.4.0 backend: " + repr(L.opj_version())

    def enc(img, cr):
        out = ctypes.c_void_p()
        n = lib.orc_opj_encode(img.ctypes.data, img.shape[0], img.shape[1], ctypes.c_float(cr), ctypes.byref(out))
        s = ctypes.string_at(out.value, n)
        lib.orc_free(out)
        return s

    def dec(s):
        b = ctypes.create_string_buffer(s, len(s))
        out, h, w = ctypes.c_void_p(), ctypes.c_size_t(), ctypes.c_size_t()
        n = lib.orc_opj_decode(b, len(s), ctypes.byref(out), ctypes.byref(h), ctypes.byref(w))
        a = np.frombuffer(ctypes.string_at(out.value, 4 * n), np.int32).copy()
        lib.orc_free(out)
        return a

    arrays, cases = {}, []
    stored = {"nat_48x80": (48, 80, 11), "nat_96x160": (96, 160, 12)}
    for name, (h, w, seed) in stored.items():
        f = L.era5_like(h, w, seed, 1.2, 1.5)
        arrays[name] = np.ascontiguousarray((((f - f.min()) / (f.max() - f.min())) * np.float32(65535)).astype(np.uint16))
    inputs = [("formula", (32, 32, 0)), ("formula", (45, 70, 1)), ("formula", (64, 96, 2)), ("formula", (100, 130, 3)), ("formula", (181, 360, 4)),
              ("formula", (721, 1440, 5)), ("stored", "nat_48x80"), ("stored", "nat_96x160")]
    for kind, spec in inputs:
        img = formula_u16(*spec) if kind == "formula" else arrays[spec]
        big = img.size > 20000
        for cr in ((30.0, 100.0) if img.size > 500000 else (1.0, 4.0, 17.0, 60.0, 900.0)):
            s = enc(img, cr)
            c = {"input": kind, "spec": spec, "h": int(img.shape[0]), "w": int(img.shape[1]), "cr": cr, "n": len(s),
                 "stream_sha256": hashlib.sha256(s).hexdigest(), "decoded_sha256": hashlib.sha256(dec(s).tobytes()).hexdigest()}
            if not big:
                c["stream_hex"] = s.hex()
            cases.append(c)
    np.savez_compressed(os.path.join(OUT, "j2k_inputs.npz"), **arrays)
    json.dump({"openjpeg": L.opj_version(), "cases": cases}, open(os.path.join(OUT, "j2k_openjpeg.json"), "w"), indent=0)
    print(len(cases), "cases,", sum(len(c.get("stream_hex", "")) // 2 for c in cases), "stream bytes stored")


if __name__ == "__main__":
    main()
