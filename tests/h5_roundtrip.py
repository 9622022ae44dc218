"""Run under an interpreter that has h5py (here /opt/conda/bin/python3.9) with HDF5_PLUGIN_PATH=<repo>/ebcc_amd:
the real HDF5 filter pipeline (filter id 308 loaded from this build) and the direct-chunk batch path against each
other.  Prints 'OK' lines; tests/test_hdf5_gpu.py drives it."""
import os
import sys

import h5py
import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ebcc_amd import EBCC_Filter, h5_batch  # noqa: E402

out = sys.argv[1]
H, W, N = 96, 160, 6
rng = np.random.default_rng(5)
y, x = np.mgrid[0:H, 0:W]
data = np.stack([(280 + 10 * np.sin(x / (9.0 + k)) * np.cos(y / (7.0 + k)) + rng.normal(0, 0.4, (H, W))).astype(np.float32)
                 for k in range(N)])
data[3] = 1.5                                                    # constant field
opt = ("max_error_target", 0.1)

# (1) through the HDF5 filter callback, one chunk per call
with h5py.File(os.path.join(out, "cb.h5"), "w") as f:
    f.create_dataset("t", data=data, **EBCC_Filter(base_cr=20, height=H, width=W, residual_opt=opt, data_dim=3))
with h5py.File(os.path.join(out, "cb.h5"), "r") as f:
    back = f["t"][...]
    raw_cb = [f["t"].id.read_direct_chunk((k, 0, 0))[1] for k in range(N)]
assert np.abs(back - data).max() <= 0.1 * 1.01 + 1e-4, np.abs(back - data).max()
print("OK callback round trip, max error", float(np.abs(back - data).max()))

# (2) direct-chunk batch path: same bytes, readable through the callback and in batch
with h5py.File(os.path.join(out, "dc.h5"), "w") as f:
    d = h5_batch.create_dataset(f, "t", data.shape, 20, opt)
    h5_batch.write_frames(d, data, 20, opt)
with h5py.File(os.path.join(out, "dc.h5"), "r") as f:
    raw_dc = [f["t"].id.read_direct_chunk((k, 0, 0))[1] for k in range(N)]
    via_callback = f["t"][...]
    via_batch = h5_batch.read_frames(f["t"])
assert raw_dc == raw_cb, [len(a) - len(b) for a, b in zip(raw_dc, raw_cb)]
print("OK direct-chunk bytes == callback bytes", sum(len(r) for r in raw_dc))
assert np.array_equal(via_callback, back) and np.array_equal(via_batch, back)
print("OK batch read == callback read")
# (2b) the same through the C entry points (ebcc_h5_write_frames / ebcc_h5_read_frames: what a netCDF-C / CDO-style caller
#      uses; they find H5Dwrite_chunk / H5Dread_chunk in the HDF5 library h5py has loaded), on a dataset with two leading
#      dimensions, written in two calls
import ctypes  # noqa: E402
import ebcc_amd  # noqa: E402
lib = ctypes.CDLL(ebcc_amd.EBCC_FILTER_PATH)
lib.ebcc_h5_write_frames.argtypes = [ctypes.c_longlong, ctypes.c_size_t, ctypes.c_size_t, ctypes.c_void_p]
lib.ebcc_h5_read_frames.argtypes = [ctypes.c_longlong, ctypes.c_size_t, ctypes.c_size_t, ctypes.c_void_p]
data4 = np.ascontiguousarray(data.reshape(2, 3, H, W))
with h5py.File(os.path.join(out, "c.h5"), "w") as f:
    d = h5_batch.create_dataset(f, "t", data4.shape, 20, opt)
    assert lib.ebcc_h5_write_frames(d.id.id, 0, 4, data4.ctypes.data) == 0
    assert lib.ebcc_h5_write_frames(d.id.id, 4, 2, data4.reshape(N, H, W)[4:].ctypes.data) == 0
    assert lib.ebcc_h5_write_frames(d.id.id, 5, 2, data4.ctypes.data) != 0          # (past the end: refused)
with h5py.File(os.path.join(out, "c.h5"), "r") as f:
    raw_c = [f["t"].id.read_direct_chunk((k // 3, k % 3, 0, 0))[1] for k in range(N)]
    via_cb4 = f["t"][...]
    got = np.full((N, H, W), -1.0, np.float32)
    ds = f["t"]                                                  # (the hid_t lives as long as this object)
    assert lib.ebcc_h5_read_frames(ds.id.id, 0, N, got.ctypes.data) == 0
    part = np.full((2, H, W), -1.0, np.float32)
    assert lib.ebcc_h5_read_frames(ds.id.id, 3, 2, part.ctypes.data) == 0
assert raw_c == raw_cb
assert np.array_equal(via_cb4.reshape(N, H, W), back) and np.array_equal(got, back) and np.array_equal(part, back[3:5])
print("OK C direct-chunk entry points: same chunk bytes, same frames")
# (3) chunks of two frames: the filter callback receives both frames at once (one multi-tile codestream per chunk)
h2, w2 = 45, 64                          # (not a multiple of 32: the second tile of a chunk has its own JPEG 2000 geometry)
multi = np.stack([(260 + 8 * np.sin(x[:h2, :w2] / (5.0 + k)) + rng.normal(0, 0.2, (h2, w2))).astype(np.float32) for k in range(4)])
kw = dict(EBCC_Filter(base_cr=10, height=h2, width=w2, residual_opt=("max_error_target", 0.05), data_dim=3))
kw["chunks"] = (2, h2, w2)
with h5py.File(os.path.join(out, "mf.h5"), "w") as f:
    f.create_dataset("t", data=multi, **kw)
with h5py.File(os.path.join(out, "mf.h5"), "r") as f:
    back2 = f["t"][...]
    raw2 = [f["t"].id.read_direct_chunk((k, 0, 0))[1] for k in (0, 2)]
assert np.abs(back2 - multi).max() <= 0.05 * 1.01 + 1e-4, np.abs(back2 - multi).max()
print("OK two-frame chunks round trip, max error", float(np.abs(back2 - multi).max()))
np.save(os.path.join(out, "chunks2.npy"), np.array([np.frombuffer(r, np.uint8) for r in raw2], dtype=object), allow_pickle=True)
np.save(os.path.join(out, "data2.npy"), multi)
np.save(os.path.join(out, "chunks.npy"), np.array([np.frombuffer(r, np.uint8) for r in raw_cb], dtype=object), allow_pickle=True)
np.save(os.path.join(out, "data.npy"), data)
