"""GPU box, under an interpreter with h5py (/opt/conda/bin/python3.9) and HDF5_PLUGIN_PATH=<repo>/ebcc_amd:
HDF5 write / read rates of 721 x 1440 frames through (a) the filter callback, one chunk per call, and
(b) the direct-chunk batch path (ebcc_amd/h5_batch.py).    python3.9 tools/gpu/h5_rate.py <dir> [frames]"""
import os
import sys
import time

os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import h5py  # noqa: E402
import numpy as np  # noqa: E402

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from ebcc_amd import EBCC_Filter, h5_batch  # noqa: E402

out, N = sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 64
as_json = "--json" in sys.argv
res = {}
H, W = 721, 1440
rng = np.random.default_rng(11)
y, x = np.mgrid[0:H, 0:W]
base = (275 + 12 * np.sin(x / 90.0) * np.cos(y / 70.0)).astype(np.float32)
data = np.stack([base + rng.normal(0, 0.8, (H, W)).astype(np.float32).cumsum(axis=1) / 20 + 0.3 * k for k in range(N)]).astype(np.float32)
opt = ("max_error_target", 0.5)
gb = data.nbytes / 1e9

ncb = min(N, 16)
# (the process's first chunk makes the engine and starts the HIP runtime - a third of a second, once: not part of the rate of
#  a writer that sends chunk after chunk)
with h5py.File(os.path.join(out, "warm.h5"), "w") as f:
    f.create_dataset("t", data=data[:2], **EBCC_Filter(base_cr=30, height=H, width=W, residual_opt=opt, data_dim=3))
with h5py.File(os.path.join(out, "warm.h5"), "r") as f:
    f["t"][...]
t0 = time.perf_counter()
with h5py.File(os.path.join(out, "cb.h5"), "w") as f:
    f.create_dataset("t", data=data[:ncb], **EBCC_Filter(base_cr=30, height=H, width=W, residual_opt=opt, data_dim=3))
t1 = time.perf_counter()
with h5py.File(os.path.join(out, "cb.h5"), "r") as f:
    back = f["t"][...]
t2 = time.perf_counter()
res["filter_callback"] = {"frames": ncb, "write_GBps": round(data[:ncb].nbytes / 1e9 / (t1 - t0), 4), "read_GBps": round(data[:ncb].nbytes / 1e9 / (t2 - t1), 4),
                          "max_abs_error": round(float(np.abs(back - data[:ncb]).max()), 5)}
print(f"filter callback ({ncb} frames): write {data[:ncb].nbytes / 1e9 / (t1 - t0):.3f} GB/s, read {data[:ncb].nbytes / 1e9 / (t2 - t1):.3f} GB/s, "
      f"max error {float(np.abs(back - data[:ncb]).max()):.4f}", flush=True, file=sys.stderr if as_json else sys.stdout)

for rep in range(2):
    back = None                                  # (the previous repetition's result - 1 GB to unmap - is not part of what is timed)
    t0 = time.perf_counter()
    with h5py.File(os.path.join(out, "dc.h5"), "w") as f:
        d = h5_batch.create_dataset(f, "t", data.shape, base_cr=30, residual_opt=opt)
        h5_batch.write_frames(d, data, 30, opt)
    t1 = time.perf_counter()
    with h5py.File(os.path.join(out, "dc.h5"), "r") as f:
        back = h5_batch.read_frames(f["t"])
    t2 = time.perf_counter()
    res["direct_chunk_batch"] = {"frames": N, "write_GBps": round(gb / (t1 - t0), 4), "read_GBps": round(gb / (t2 - t1), 4),
                                 "max_abs_error": round(float(np.abs(back - data).max()), 5), "file_MB": round(os.path.getsize(os.path.join(out, "dc.h5")) / 1e6, 2)}
    print(f"direct-chunk batch ({N} frames), rep {rep}: write {gb / (t1 - t0):.3f} GB/s, read {gb / (t2 - t1):.3f} GB/s, "
          f"max error {float(np.abs(back - data).max()):.4f}, file {os.path.getsize(os.path.join(out, 'dc.h5')) / 1e6:.1f} MB", flush=True, file=sys.stderr if as_json else sys.stdout)
# the same through the C entry points (ebcc_h5_write_frames / ebcc_h5_read_frames, include/ebcc_hip.h)
import ctypes  # noqa: E402
import ebcc_amd  # noqa: E402
clib = ctypes.CDLL(ebcc_amd.EBCC_FILTER_PATH)
clib.ebcc_h5_write_frames.argtypes = [ctypes.c_longlong, ctypes.c_size_t, ctypes.c_size_t, ctypes.c_void_p]
clib.ebcc_h5_read_frames.argtypes = [ctypes.c_longlong, ctypes.c_size_t, ctypes.c_size_t, ctypes.c_void_p]
for rep in range(2):
    back = None
    t0 = time.perf_counter()
    with h5py.File(os.path.join(out, "c.h5"), "w") as f:
        d = h5_batch.create_dataset(f, "t", data.shape, base_cr=30, residual_opt=opt)
        assert clib.ebcc_h5_write_frames(d.id.id, 0, N, data.ctypes.data) == 0
    t1 = time.perf_counter()
    with h5py.File(os.path.join(out, "c.h5"), "r") as f:
        d = f["t"]
        back = np.empty_like(data)
        assert clib.ebcc_h5_read_frames(d.id.id, 0, N, back.ctypes.data) == 0
    t2 = time.perf_counter()
    res["c_direct_chunk"] = {"frames": N, "write_GBps": round(gb / (t1 - t0), 4), "read_GBps": round(gb / (t2 - t1), 4),
                             "max_abs_error": round(float(np.abs(back - data).max()), 5)}
    print(f"C direct-chunk entry points ({N} frames), rep {rep}: write {gb / (t1 - t0):.3f} GB/s, read {gb / (t2 - t1):.3f} GB/s", flush=True, file=sys.stderr if as_json else sys.stdout)
# a dataset of several device batches: the batches alternate between two engine sets (host part of one beside the kernels of
# the next; one batch downloaded while the next decodes), the next call's chunks are fetched meanwhile
M = 4 * N
if "--big" in sys.argv:
    big = np.concatenate([data + np.float32(0.11 * r) for r in range(M // N)])
    for rep in range(2):
        back = None
        t0 = time.perf_counter()
        with h5py.File(os.path.join(out, "big.h5"), "w") as f:
            d = h5_batch.create_dataset(f, "t", big.shape, base_cr=30, residual_opt=opt)
            h5_batch.write_frames(d, big, 30, opt)
        t1 = time.perf_counter()
        with h5py.File(os.path.join(out, "big.h5"), "r") as f:
            back = h5_batch.read_frames(f["t"])
        t2 = time.perf_counter()
        res["direct_chunk_batches_x4"] = {"frames": M, "write_GBps": round(big.nbytes / 1e9 / (t1 - t0), 4), "read_GBps": round(big.nbytes / 1e9 / (t2 - t1), 4),
                                          "max_abs_error": round(float(np.abs(back[::5] - big[::5]).max()), 5)}
        print(f"direct-chunk batches ({M} frames), rep {rep}: write {big.nbytes / 1e9 / (t1 - t0):.3f} GB/s, read {big.nbytes / 1e9 / (t2 - t1):.3f} GB/s",
              flush=True, file=sys.stderr if as_json else sys.stdout)
if as_json:
    import json
    print(json.dumps(res), flush=True)
