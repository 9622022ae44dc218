"""GPU box: many steps of the default workload in one process - host RSS, device memory in use and the step time every 25 steps
(leaks, creeping slow-downs, rare faults).   python tools/gpu/soak.py [steps]"""
import ctypes
import os
import sys
import time

os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402
import bench  # noqa: E402
from tests import _lib as L  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
n, H, W = 256, 721, 1440
lib = L.product()
device = torch.device("cuda:0")
frames = bench.synth_frames(torch, n, device, seed=0)
out = torch.empty_like(frames)
cfg = L.make_config((1, H, W), base_cr=30.0, error=0.5, residual_type=L.MAX_ERROR)
outs = (ctypes.c_void_p * n)()
sizes = (ctypes.c_size_t * n)()


def rss_mb():
    for line in open("/proc/self/status"):
        if line.startswith("VmRSS"):
            return int(line.split()[1]) / 1024.0
    return 0.0


with L.Context(n, H, W) as ctx:
    ptr = ctypes.c_void_p(ctx.ptr)
    first = None
    t_acc = 0.0
    for i in range(steps):
        t0 = time.perf_counter()
        assert lib.ebcc_hip_encode_frames(ptr, ctypes.c_void_p(frames.data_ptr()), n, ctypes.byref(cfg), outs, sizes) == 0
        assert lib.ebcc_hip_decode_frames(ptr, outs, sizes, n, ctypes.c_void_p(out.data_ptr())) == 0
        torch.cuda.synchronize()
        t_acc += time.perf_counter() - t0
        total = sum(sizes[k] for k in range(n))
        if first is None:
            first = total
        assert total == first, (i, total, first)
        for k in range(n):
            lib.free_buffer(ctypes.c_void_p(outs[k]))
        if (i + 1) % 25 == 0:
            free, tot = torch.cuda.mem_get_info()
            print(f"step {i + 1}: {t_acc / 25 * 1e3:.1f} ms per step, host RSS {rss_mb():.0f} MB, device memory in use {(tot - free) / 1e9:.2f} GB, max error {float((out - frames).abs().max()):.4f}", flush=True)
            t_acc = 0.0
