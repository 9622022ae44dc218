"""GPU box: rate of the reference-compatible host-pointer entry points (PCIe inclusive): ebcc_encode_chunking /
ebcc_decode_chunking on a host array of N x 721 x 1440 fp32 with one frame per chunk.
    python tools/gpu/host_api_rate.py [frames]"""
import ctypes
import os
import sys
import time

os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import numpy as np  # noqa: E402

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tests import _lib as L  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
lib = L.product()
x = np.stack([L.era5_like(721, 1440, s) for s in range(8)] * (n // 8)).astype(np.float32)
x += np.arange(n, dtype=np.float32)[:, None, None] * np.float32(0.37)
cfg = L.make_config(x.shape, (1, 721, 1440), base_cr=30.0, error=0.5, residual_type=L.MAX_ERROR)
for rep in range(3):
    out = ctypes.c_void_p()
    t0 = time.perf_counter()
    nb = lib.ebcc_encode_chunking(x.ctypes.data, ctypes.byref(cfg), ctypes.byref(out))
    t1 = time.perf_counter()
    assert nb > 0
    dec = ctypes.c_void_p()
    m = lib.ebcc_decode_chunking(ctypes.c_void_p(out.value), nb, ctypes.byref(dec))
    t2 = time.perf_counter()
    assert m == x.size
    if rep == 2:
        d = np.ctypeslib.as_array(ctypes.cast(dec, ctypes.POINTER(ctypes.c_float)), shape=x.shape)
        print("max abs error", float(np.abs(d - x).max()), "ratio", x.nbytes / nb)
    lib.free_buffer(out)
    lib.free_buffer(dec)
    print(f"rep {rep}: encode {x.nbytes / (t1 - t0) / 1e9:.3f} GB/s  decode {x.nbytes / (t2 - t1) / 1e9:.3f} GB/s  "
          f"round trip {x.nbytes / (t2 - t0) / 1e9:.3f} GB/s  ({(t2 - t0) * 1e3:.0f} ms for {n} frames)", flush=True)
