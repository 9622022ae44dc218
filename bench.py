#!/usr/bin/env python3
"""bench.py - EBCC per-frame hot path on MI355X: encode + decode of a batch of synthetic ERA5-shaped frames.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A "step" = ebcc_encode of every frame of the batch (inputs resident in HBM; compressed streams land in host
memory, zstd on host cores as in the reference) followed by ebcc_decode of those streams back into HBM.
Workload = BASELINE.json configs[1]: 256 frames 721x1440 fp32, base_cr=30, MAX_ERROR=0.5 per GPU (weak
scaling: frames are independent, every rank codes its own batch, no collective on the data path).
Rank 0 prints ONE JSON line.  `value` = frames * 4 152 960 B / step time summed over ranks (round trip).
"""
import argparse
import ctypes
import json
import os
import sys
import time

# The encode path runs its batch as 4 concurrent slices when the HIP runtime has a hardware queue for each of their
# streams (host_codec.hip: default_encode_slices); the runtime reads this when it starts, i.e. before torch loads.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

import numpy as np  # noqa: E402

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

H, W = 721, 1440
FRAME_BYTES = H * W * 4
BASE_CR, MAX_ERR = 30.0, 0.5


def synth_frames(torch, n, device, seed):
    """SURVEY.md section 8(d) generator on the device: k^-1.5 spectrum noise on a zonal profile."""
    g = torch.Generator(device=device)
    g.manual_seed(1234 + seed)
    ky = torch.fft.fftfreq(H, device=device)[:, None]
    kx = torch.fft.rfftfreq(W, device=device)[None, :]
    k = torch.sqrt(ky * ky + kx * kx)
    k[0, 0] = 1
    filt = k ** (-1.5)
    filt[0, 0] = 0
    lat = torch.linspace(-1, 1, H, device=device)[:, None]
    prof = 235 + 50 * torch.cos(lat * torch.pi / 2)
    out = torch.empty((n, H, W), dtype=torch.float32, device=device)
    for i in range(0, n, 32):
        m = min(32, n - i)
        noise = torch.randn((m, H, W), generator=g, device=device, dtype=torch.float32)
        f = torch.fft.irfft2(torch.fft.rfft2(noise) * filt, s=(H, W))
        f = f / f.std(dim=(1, 2), keepdim=True)
        out[i:i + m] = (prof + 2.5 * f).to(torch.float32)
    return out.contiguous()


def cpu_baseline(sample, cores):
    """Reference CPU codec (oracle/_ref = reference sources + OpenJPEG 2.4.0/zstd) or, if that build cannot
    be loaded, the oracle port; one process per core, one frame each."""
    import multiprocessing as mp
    from tests import _lib as L
    kind = "reference" if os.path.exists(L.REF_SO) else "port"
    try:
        if kind == "reference":
            ctypes.CDLL(L.REF_SO)
    except OSError:
        kind = "port"
    frames = [np.ascontiguousarray(sample[i % len(sample)]) for i in range(cores)]
    t0 = time.time()
    with mp.get_context("fork").Pool(cores) as pool:
        res = pool.map(_cpu_one, [(kind, f) for f in frames])
    wall = time.time() - t0
    enc = float(np.mean([r[0] for r in res]))
    dec = float(np.mean([r[1] for r in res]))
    return {"value": round(len(frames) * FRAME_BYTES / wall / 1e9, 6), "unit": "GB/s", "cores": cores, "kind": kind,
            "sample": f"{len(frames)} frames 721x1440, one per process, encode+decode; mean {enc:.2f}s enc / {dec:.3f}s dec per frame",
            "encode_MBps_per_core": round(FRAME_BYTES / enc / 1e6, 3), "decode_MBps_per_core": round(FRAME_BYTES / dec / 1e6, 2)}


def _cpu_one(arg):
    kind, frame = arg
    from tests import _lib as L
    cfg = L.make_config((1, H, W), base_cr=BASE_CR, error=MAX_ERR, residual_type=L.MAX_ERROR)
    if kind == "reference":
        lib = ctypes.CDLL(L.REF_SO)
        lib.ebcc_encode.restype = ctypes.c_size_t
        lib.ebcc_encode.argtypes = [ctypes.c_void_p, ctypes.POINTER(L.CodecConfig), L.c_void_pp]
        lib.ebcc_decode.restype = ctypes.c_size_t
        lib.ebcc_decode.argtypes = [ctypes.c_void_p, ctypes.c_size_t, L.c_void_pp]
        out = ctypes.c_void_p()
        t = time.time()
        n = lib.ebcc_encode(frame.ctypes.data, ctypes.byref(cfg), ctypes.byref(out))
        te = time.time() - t
        dec = ctypes.c_void_p()
        t = time.time()
        lib.ebcc_decode(out, n, ctypes.byref(dec))
        td = time.time() - t
    else:
        t = time.time()
        s = L.orc_encode(frame, cfg)
        te = time.time() - t
        t = time.time()
        L.orc_decode(s)
        td = time.time() - t
    return te, td


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--frames", type=int, default=256, help="frames per GPU per step")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: no HIP device visible (the codec has no CPU fallback)")
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if world > 1:
        dist.init_process_group("nccl", device_id=device)

    from tests import _lib as L                     # ctypes bindings of the C-ABI (after torch: one HIP runtime)
    lib = L.product()
    lib.ebcc_hip_timing_enable.argtypes = [ctypes.c_void_p, ctypes.c_int]
    lib.ebcc_hip_timing_read.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.POINTER(ctypes.c_double),
                                         ctypes.POINTER(ctypes.c_long)]
    n = args.frames
    ctx = lib.ebcc_hip_create(local_rank, n, H, W)
    assert ctx, lib.ebcc_hip_last_error()
    lib.ebcc_hip_prepare.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
    assert lib.ebcc_hip_prepare(ctx, n) == 0                    # slice engines: part of the context, not of a step
    frames = synth_frames(torch, n, device, seed=rank)
    out = torch.empty_like(frames)
    torch.cuda.synchronize()
    cfg = L.make_config((1, H, W), base_cr=BASE_CR, error=MAX_ERR, residual_type=L.MAX_ERROR)
    outs = (ctypes.c_void_p * n)()
    sizes = (ctypes.c_size_t * n)()

    def step():
        t0 = time.perf_counter()
        rc = lib.ebcc_hip_encode_frames(ctx, frames.data_ptr(), n, ctypes.byref(cfg), outs, sizes)
        assert rc == 0, lib.ebcc_hip_last_error()
        t1 = time.perf_counter()
        rc = lib.ebcc_hip_decode_frames(ctx, outs, sizes, n, out.data_ptr())
        assert rc == 0, lib.ebcc_hip_last_error()
        t2 = time.perf_counter()
        nbytes = sum(sizes[i] for i in range(n))
        for i in range(n):
            lib.free_buffer(outs[i])
        return t1 - t0, t2 - t1, nbytes

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    lib.ebcc_hip_timing_enable(ctx, 1)
    barrier()
    t0 = time.perf_counter()
    enc_t = dec_t = 0.0
    comp = 0
    for _ in range(args.steps):
        e, d, comp = step()
        enc_t += e
        dec_t += d
    barrier()
    elapsed = time.perf_counter() - t0
    lib.ebcc_hip_timing_enable(ctx, 0)

    # parity guard on the timed data: the error bound holds on every frame (size-independent property)
    max_err = float((out - frames).abs().amax())
    assert max_err <= MAX_ERR * 1.01 + 1e-3, max_err

    t = torch.tensor([elapsed], dtype=torch.float64, device=device)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())

    if rank == 0:
        ms_step = elapsed / args.steps * 1e3
        total_frames = n * world
        value = total_frames * FRAME_BYTES / (elapsed / args.steps) / 1e9
        # dominant kernel by total time: tier-1 coding of every code-block (HIP events recorded on the launching
        # stream of each slice engine around both of its phases)
        tms, launches = ctypes.c_double(), ctypes.c_long()
        lib.ebcc_hip_timing_read(ctx, b"t1_encode", ctypes.byref(tms), ctypes.byref(launches))
        kern = {}
        for name in (b"t1_encode", b"t1_symbols", b"t1_mq", b"t1_probe_decode", b"rate_alloc", b"j2k_dwt_fwd", b"spiht_encode", b"t1_decode"):
            a, c = ctypes.c_double(), ctypes.c_long()
            lib.ebcc_hip_timing_read(ctx, name, ctypes.byref(a), ctypes.byref(c))
            if c.value:
                kern[name.decode()] = {"ms_avg": round(a.value / c.value, 4), "launches": c.value}
        def pmc_traffic(frames_per_launch):
            """HBM bytes per launch of the tier-1 encoder from the committed PMC passes (profiles/r01_pmc_*.json,
            tools/gpu/profile.sh: separate FETCH_SIZE and WRITE_SIZE runs of `--frames 64` on one slice = 64 frames per
            dispatch).  gfx950 correction of the micro-architecture guide: FETCH_SIZE x 2; units of 1 KB."""
            try:
                fj = json.load(open(os.path.join(ROOT, "profiles", "r01_pmc_fetch_size.json")))
                wj = json.load(open(os.path.join(ROOT, "profiles", "r01_pmc_write_size.json")))
                f, w, per = fj["kernels"], wj["kernels"], float(fj.get("frames_per_dispatch") or 32)
                per_frame = sum(2.0 * f[k]["per_dispatch"] + w[k]["per_dispatch"] for k in ("k_t1_symbols", "k_t1_mq")) * 1024.0 / per
                return int(per_frame * frames_per_launch)
            except Exception:
                return None

        roof = None
        if launches.value:
            avg_s = tms.value / launches.value / 1e3
            # fp32 read + compressed bytes written per launch (a batch runs as several concurrent slices,
            # each with its own launch)
            algo = (n * FRAME_BYTES + comp) * args.steps / launches.value
            ach = algo / avg_s / 1e9
            roof = {"bound": "hbm", "kernel": "tier-1 encoder (k_t1_symbols + k_t1_mq)", "achieved": round(ach, 3), "peak": 8000.0, "unit": "GB/s",
                    "frac": round(ach / 8000.0, 6), "traffic": pmc_traffic(n * args.steps / launches.value), "avg_launch_ms": round(avg_s * 1e3, 4),
                    "algorithmic_bytes_per_launch": algo}
        line = {
            "metric": "fp32 GB/s encode+decode, 721x1440 ERA5 frames MAX_ERROR=0.5",
            "value": round(value, 4), "unit": "GB/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_step, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{n}-frame batch 721x1440 fp32 per GPU, base_cr=30 MAX_ERROR=0.5 (BASELINE configs[1])",
                       "frames_per_gpu": n, "parallelism": f"frames sharded over {world} GPU(s), no collective",
                       "hw_queues": int(os.environ.get("GPU_MAX_HW_QUEUES", "4")),
                       "encode_slices": int(os.environ.get("EBCC_HIP_SLICES", "4" if int(os.environ.get("GPU_MAX_HW_QUEUES", "4")) >= 8 else "2"))},
            "encode_GBps": round(total_frames * FRAME_BYTES * args.steps / enc_t / 1e9, 4),
            "decode_GBps": round(total_frames * FRAME_BYTES * args.steps / dec_t / 1e9, 4),
            "compressed_bytes_per_frame": int(comp / n), "max_abs_error": round(max_err, 5),
            "kernels": kern, "roofline": roof,
        }
        if not args.no_cpu_baseline:
            try:
                cores = min(16, os.cpu_count() or 1)
                line["cpu_baseline"] = cpu_baseline(frames[:4].cpu().numpy(), cores)
            except Exception as e:                              # the baseline is a report, never a gate
                line["cpu_baseline"] = {"error": repr(e)}
        print(json.dumps(line), flush=True)
    lib.ebcc_hip_destroy(ctx)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
