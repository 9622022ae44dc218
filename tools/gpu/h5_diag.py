"""GPU box, conda python: where the time of h5_batch.read_frames goes (fetch / decode / download)."""
import os, sys, time, ctypes
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import h5py, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from ebcc_amd import h5_batch
out, N = sys.argv[1], int(sys.argv[2])
H, W = 721, 1440
rng = np.random.default_rng(11)
y, x = np.mgrid[0:H, 0:W]
base = (275 + 12 * np.sin(x / 90.0) * np.cos(y / 70.0)).astype(np.float32)
data = np.stack([base + 0.3 * k for k in range(N)]).astype(np.float32)
opt = ("max_error_target", 0.5)
with h5py.File(os.path.join(out, "d.h5"), "w") as f:
    d = h5_batch.create_dataset(f, "t", data.shape, base_cr=30, residual_opt=opt)
    h5_batch.write_frames(d, data, 30, opt)
with h5py.File(os.path.join(out, "d.h5"), "r") as f:
    ds = f["t"]
    t0 = time.perf_counter(); codec = h5_batch.BatchCodec(H, W, N); t1 = time.perf_counter()
    raw = [ds.id.read_direct_chunk((i, 0, 0))[1] for i in range(N)]; t2 = time.perf_counter()
    o = np.empty((N, H, W), np.float32); t3 = time.perf_counter()
    for rep in range(2):
        ta = time.perf_counter()
        n = len(raw)
        ptrs = (ctypes.c_void_p * n)(*[ctypes.cast(ctypes.c_char_p(s), ctypes.c_void_p).value for s in raw])
        sizes = (ctypes.c_size_t * n)(*[len(s) for s in raw])
        tb = time.perf_counter()
        rc = codec.lib.ebcc_hip_decode_frames(codec.ctx, ptrs, sizes, n, codec.d_buf); tc = time.perf_counter()
        rc2 = codec.lib.ebcc_hip_download(codec.ctx, o.ctypes.data, codec.d_buf, o.nbytes); td = time.perf_counter()
        print(f"rep {rep}: pointers {tb-ta:.3f}s decode {tc-tb:.3f}s (rc {rc}) download {td-tc:.3f}s (rc {rc2})", flush=True)
    print(f"engine {t1-t0:.3f}s, {N} read_direct_chunk {t2-t1:.3f}s, np.empty {t3-t2:.3f}s; max err {float(np.abs(o-data).max()):.4f}")
    t4 = time.perf_counter(); back = h5_batch.read_frames(ds, codec=codec); t5 = time.perf_counter()
    print(f"read_frames with a live codec: {t5-t4:.3f}s = {data.nbytes/1e9/(t5-t4):.2f} GB/s")
    codec.close()
    t4 = time.perf_counter(); back = h5_batch.read_frames(ds); t5 = time.perf_counter()
    print(f"read_frames (own codec): {t5-t4:.3f}s = {data.nbytes/1e9/(t5-t4):.2f} GB/s")
