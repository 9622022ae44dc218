"""GPU box: the 4096-frame RELATIVE_ERROR workload of bench.py (config 3) several times in one process - how much of its run-to-run
spread is inside a process?   python tools/gpu/config3_reps.py [frames] [reps]"""
import ctypes
import os
import sys
import time

os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402
import bench  # noqa: E402
from tests import _lib as L  # noqa: E402

m = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 4
n, H, W = 256, 721, 1440
lib = L.product()
device = torch.device("cuda:0")
data = bench.synth_frames(torch, m, device, seed=77, ramp=(0.25, 1.75))
dec = torch.empty_like(data)
cfg = L.make_config((1, H, W), base_cr=30.0, error=1e-3, residual_type=L.RELATIVE_ERROR)
outs = (ctypes.c_void_p * m)()
sizes = (ctypes.c_size_t * m)()
lib.ebcc_hip_encode_shard.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
lib.ebcc_hip_decode_shard.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]
with L.Context(n, H, W) as ctx:
    ptr = ctypes.c_void_p(ctx.ptr)
    for r in range(reps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        assert lib.ebcc_hip_encode_shard(ptr, ctypes.c_void_p(data.data_ptr()), m, ctypes.byref(cfg), outs, sizes) == 0
        t1 = time.perf_counter()
        assert lib.ebcc_hip_decode_shard(ptr, outs, sizes, m, ctypes.c_void_p(dec.data_ptr())) == 0
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        t3 = time.perf_counter()
        for i in range(m):
            lib.free_buffer(ctypes.c_void_p(outs[i]))
        t4 = time.perf_counter()
        raw = m * H * W * 4 / 1e9
        print(f"rep {r}: encode {raw / (t1 - t0):.2f} GB/s ({(t1 - t0) * 1e3 / (m / n):.1f} ms per batch), decode {raw / (t2 - t1):.2f} GB/s, freeing the streams {(t4 - t3) * 1e3:.0f} ms", flush=True)
