"""GPU box: latency of ONE frame through the reference's per-chunk entry points (what an HDF5 filter callback pays)."""
import ctypes
import os
import sys
import time

os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import numpy as np  # noqa: E402

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tests import _lib as L  # noqa: E402

lib = L.product()
for h, w in ((721, 1440), (181, 360)):
    x = L.era5_like(h, w, 3)
    cfg = L.make_config((1, h, w), base_cr=30.0, error=0.5, residual_type=L.MAX_ERROR)
    te, td = [], []
    for rep in range(6):
        out = ctypes.c_void_p()
        t0 = time.perf_counter()
        nb = lib.ebcc_encode(x.ctypes.data, ctypes.byref(cfg), ctypes.byref(out))
        t1 = time.perf_counter()
        dec = ctypes.c_void_p()
        m = lib.ebcc_decode(ctypes.c_void_p(out.value), nb, ctypes.byref(dec))
        t2 = time.perf_counter()
        lib.free_buffer(out); lib.free_buffer(dec)
        te.append(t1 - t0); td.append(t2 - t1)
    print(f"{h}x{w}: ebcc_encode {min(te[1:]) * 1e3:.1f} ms, ebcc_decode {min(td[1:]) * 1e3:.1f} ms (first call {te[0] * 1e3:.0f} / {td[0] * 1e3:.0f} ms), {nb} bytes", flush=True)
