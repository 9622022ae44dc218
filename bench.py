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
import hashlib
import json
import os
import sys
import time

# The encode path runs its batch as concurrent slices (three by default, host_codec.hip: default_encode_slices), each with two
# streams; the HIP runtime has a hardware queue for each when GPU_MAX_HW_QUEUES >= 8 - it reads this when it starts, i.e.
# before torch loads.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

import numpy as np  # noqa: E402

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

H, W = 721, 1440
FRAME_BYTES = H * W * 4
BASE_CR, MAX_ERR = 30.0, 0.5


def cgroup_cpu():
    """CPU quota and throttling counters of this container (cgroup v2 cpu.max / cpu.stat, or the v1 files): the GPU box
    of this project is a 16-CPU quota on a 256-thread host, and the level-22 zstd stage is ~1.3 core-seconds per step."""
    out = {"quota_cpus": None, "nr_periods": None, "nr_throttled": None, "throttled_usec": None, "usage_usec": None}
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        out["quota_cpus"] = None if q == "max" else round(int(q) / int(per), 3)
        for line in open("/sys/fs/cgroup/cpu.stat"):
            k, v = line.split()
            if k in out:
                out[k] = int(v)
        return out
    except OSError:
        pass
    for d in ("/sys/fs/cgroup/cpu", "/sys/fs/cgroup/cpu,cpuacct"):
        try:
            q = int(open(d + "/cpu.cfs_quota_us").read())
            per = int(open(d + "/cpu.cfs_period_us").read())
            out["quota_cpus"] = None if q <= 0 else round(q / per, 3)
            for line in open(d + "/cpu.stat"):
                k, v = line.split()
                if k == "throttled_time":
                    out["throttled_usec"] = int(v) // 1000
                elif k in out:
                    out[k] = int(v)
            return out
        except OSError:
            continue
    return out


def synth_frames(torch, n, device, seed, slope=1.5, amp=2.5, ramp=None):
    """SURVEY.md section 8(d) generator on the device: k^-slope spectrum noise (amplitude amp) on a zonal profile;
    ramp = (a, b): frame i's noise amplitude and profile swing are scaled by a + (b - a) * i / (n - 1), so that the
    frames' ranges differ (config 3: RELATIVE_ERROR targets follow the per-frame range)."""
    g = torch.Generator(device=device)
    g.manual_seed(1234 + seed)
    ky = torch.fft.fftfreq(H, device=device)[:, None]
    kx = torch.fft.rfftfreq(W, device=device)[None, :]
    k = torch.sqrt(ky * ky + kx * kx)
    k[0, 0] = 1
    filt = k ** (-slope)
    filt[0, 0] = 0
    lat = torch.linspace(-1, 1, H, device=device)[:, None]
    swing = 50 * torch.cos(lat * torch.pi / 2)
    prof = 235 + swing
    out = torch.empty((n, H, W), dtype=torch.float32, device=device)
    for i in range(0, n, 32):
        m = min(32, n - i)
        noise = torch.randn((m, H, W), generator=g, device=device, dtype=torch.float32)
        f = torch.fft.irfft2(torch.fft.rfft2(noise) * filt, s=(H, W))
        f = f / f.std(dim=(1, 2), keepdim=True)
        if ramp is None:
            out[i:i + m] = (prof + amp * f).to(torch.float32)
        else:
            idx = torch.arange(i, i + m, device=device, dtype=torch.float32) / max(1, n - 1)
            scale = (ramp[0] + (ramp[1] - ramp[0]) * idx)[:, None, None]
            out[i:i + m] = (235 + scale * (swing + amp * f)).to(torch.float32)
    return out.contiguous()


def cpu_baseline(sample, cores, checks=()):
    """Reference CPU codec (oracle/_ref = reference sources + OpenJPEG 2.4.0/zstd) or, if that build cannot
    be loaded, the oracle port; one process per core, two frames each, median per-frame times.
    `checks`: (tag, frame, sha256 of the MI355X stream) - these frames are among the ones the reference codes, and the
    hash of what it wrote is compared with the device's: byte parity on the very data the bench timed."""
    import hashlib
    import multiprocessing as mp
    from tests import _lib as L
    kind = "reference" if os.path.exists(L.REF_SO) else "port"
    try:
        if kind == "reference":
            ctypes.CDLL(L.REF_SO)
    except OSError:
        kind = "port"
    frames = [np.ascontiguousarray(c[1]) for c in checks]
    frames += [np.ascontiguousarray(sample[i % len(sample)]) for i in range(max(0, 2 * cores - len(frames)))]
    t0 = time.time()
    with mp.get_context("spawn").Pool(cores) as pool:      # (never fork a process that has initialised HIP)
        modes = [c[3] if len(c) > 3 else "max" for c in checks] + ["max"] * (len(frames) - len(checks))
        res = pool.map(_cpu_one, [(kind, f, m) for f, m in zip(frames, modes)], chunksize=1)
    wall = time.time() - t0
    timed = [r for r, m in zip(res, modes) if m == "max"]          # (the timing is the MAX_ERROR workload's)
    enc = float(np.median([r[0] for r in timed]))
    dec = float(np.median([r[1] for r in timed]))
    out = {"value": round(len(frames) * FRAME_BYTES / wall / 1e9, 6), "unit": "GB/s", "cores": cores, "kind": kind,
           "sample": f"{len(frames)} frames 721x1440 (base_cr 30, MAX_ERROR 0.5), two per process on {cores} processes, encode+decode; "
                     f"median {enc:.2f}s enc / {dec:.3f}s dec per frame, {wall:.1f}s wall",
           "encode_MBps_per_core": round(FRAME_BYTES / enc / 1e6, 3), "decode_MBps_per_core": round(FRAME_BYTES / dec / 1e6, 2)}
    if checks:
        same = [c[0] for c, r in zip(checks, res) if r[2] == c[2]]
        diff = [c[0] for c, r in zip(checks, res) if r[2] != c[2]]
        out["stream_parity"] = {"what": f"sha256 of the MI355X stream == sha256 of the {kind} codec's stream for the same frame",
                                "identical": same, "different": diff}
    return out


def _cpu_one(arg):
    import hashlib
    kind, frame, mode = arg
    from tests import _lib as L
    cfg = (L.make_config((1, H, W), base_cr=BASE_CR, error=MAX_ERR, residual_type=L.MAX_ERROR) if mode == "max" else
           L.make_config((1, H, W), base_cr=BASE_CR, error=1e-3, residual_type=L.RELATIVE_ERROR))
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
        digest = hashlib.sha256(ctypes.string_at(out, n)).hexdigest()
        dec = ctypes.c_void_p()
        t = time.time()
        lib.ebcc_decode(out, n, ctypes.byref(dec))
        td = time.time() - t
    else:
        t = time.time()
        s = L.orc_encode(frame, cfg)
        te = time.time() - t
        digest = hashlib.sha256(s).hexdigest()
        t = time.time()
        L.orc_decode(s)
        td = time.time() - t
    return te, td, digest


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--frames", type=int, default=256, help="frames per GPU per step")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the extra workloads (config3, residual population, host API)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: no HIP device visible (the codec has no CPU fallback)")
    # EBCC_BENCH_SHARE_GPU=1 (rehearsal on a box with fewer GPUs than ranks: tests/test_bench_ranks_gpu.py): the ranks take the
    # visible devices round-robin and talk through gloo - RCCL refuses two ranks on one device; the rank logic (barrier,
    # max-over-ranks time, gathered host figures, whole-job value) is the same
    share = os.environ.get("EBCC_BENCH_SHARE_GPU") == "1"
    dev_index = local_rank % torch.cuda.device_count() if share else local_rank
    torch.cuda.set_device(dev_index)
    device = torch.device("cuda", dev_index)
    if world > 1:
        if share:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=device)

    from tests import _lib as L                     # ctypes bindings of the C-ABI (after torch: one HIP runtime)
    lib = L.product()
    lib.ebcc_hip_timing_enable.argtypes = [ctypes.c_void_p, ctypes.c_int]
    lib.ebcc_hip_timing_read.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.POINTER(ctypes.c_double),
                                         ctypes.POINTER(ctypes.c_long)]
    n = args.frames
    ctx = lib.ebcc_hip_create(dev_index, n, H, W)
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
        if step.hash:                                            # (the last warm-up step only: not inside the timed region)
            step.first_hashes = [hashlib.sha256(ctypes.string_at(outs[i], sizes[i])).hexdigest() for i in range(min(n, 2))]
        for i in range(n):
            lib.free_buffer(outs[i])
        return t1 - t0, t2 - t1, nbytes

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    lib.ebcc_hip_host_stats.argtypes = [ctypes.POINTER(ctypes.c_double), ctypes.c_int]
    lib.ebcc_hip_host_stats.restype = None
    hstats = (ctypes.c_double * 8)()
    step.hash = False
    for w in range(max(1, args.warmup)):
        step.hash = w == max(1, args.warmup) - 1
        step()
    step.hash = False
    lib.ebcc_hip_timing_enable(ctx, 1)
    lib.ebcc_hip_host_stats(hstats, 1)                          # (reset)
    barrier()
    cg0 = cgroup_cpu()
    ru0 = os.times()
    t0 = time.perf_counter()
    enc_t = dec_t = 0.0
    comp = 0
    for _ in range(args.steps):
        e, d, comp = step()
        enc_t += e
        dec_t += d
    barrier()
    elapsed = time.perf_counter() - t0
    ru1 = os.times()
    cg1 = cgroup_cpu()
    lib.ebcc_hip_timing_enable(ctx, 0)
    lib.ebcc_hip_host_stats(hstats, 0)
    # host side of this rank over the timed region (per step): what the entropy stage cost, what the process burnt in all
    # its threads, and whether the container's CPU quota throttled it
    lib.ebcc_hip_default_encode_slices.restype = ctypes.c_int
    lib.ebcc_hip_encode_slices_for.restype = ctypes.c_int
    lib.ebcc_hip_encode_slices_for.argtypes = [ctypes.c_size_t]
    default_slices = int(lib.ebcc_hip_encode_slices_for(n))     # (what a batch of n frames runs as: the environment and the batch size decide)
    host = {"pool_threads": lib.ebcc_hip_host_threads(default_slices),
            "usable_cpus": int(hstats[0]), "quota_cpus": hstats[1] or None,
            "zstd_core_s_per_step": round(hstats[2] / args.steps, 4), "zstd_wait_ms_per_step": round(hstats[3] / args.steps * 1e3, 2),
            "zstd_MB_per_step": round(hstats[4] / args.steps / 1e6, 3),
            "prefix_MB_per_step_decided_without_zstd": round(hstats[6] / args.steps / 1e6, 3),
            "process_cpu_s_per_step": round(((ru1.user - ru0.user) + (ru1.system - ru0.system)) / args.steps, 4)}
    if cg0["nr_throttled"] is not None and cg1["nr_throttled"] is not None:
        host["cgroup"] = {"quota_cpus": cg1["quota_cpus"], "periods": cg1["nr_periods"] - cg0["nr_periods"],
                          "throttled_periods": cg1["nr_throttled"] - cg0["nr_throttled"],
                          "throttled_ms_per_step": round((cg1["throttled_usec"] - cg0["throttled_usec"]) / 1e3 / args.steps, 2),
                          "cpu_s_per_step": None if cg0["usage_usec"] is None else round((cg1["usage_usec"] - cg0["usage_usec"]) / 1e6 / args.steps, 4)}

    # parity guard on the timed data: the error bound holds on every frame (size-independent property)
    max_err = float((out - frames).abs().amax())
    assert max_err <= MAX_ERR * 1.01 + 1e-3, max_err

    t = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if share else device)
    hosts = [host]
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        hosts = [None] * world
        dist.all_gather_object(hosts, host)
    elapsed = float(t.item())

    parity_checks = []                                              # (tag, frame, sha256 of its MI355X stream) -> cpu_baseline

    def h5_path_rates(frames_):
        import subprocess
        import tempfile
        conda = "/opt/conda/bin/python3.9"
        if not os.path.exists(conda):
            return {"skipped": "no interpreter with h5py in this image"}
        env = dict(os.environ, HDF5_PLUGIN_PATH=os.path.join(ROOT, "ebcc_amd"), HDF5_USE_FILE_LOCKING="FALSE")
        env.pop("PYTHONPATH", None)
        try:
            with tempfile.TemporaryDirectory(dir="/dev/shm" if os.path.isdir("/dev/shm") else None) as tmp:
                r = subprocess.run([conda, os.path.join(ROOT, "tools", "gpu", "h5_rate.py"), tmp, str(frames_), "--json", "--big"], env=env,
                                   capture_output=True, text=True, timeout=600)
            if r.returncode != 0:
                return {"error": (r.stderr or r.stdout)[-400:]}
            out_ = json.loads(r.stdout.strip().splitlines()[-1])
            out_["workload"] = f"{frames_} one-frame chunks 721x1440 (error bound 0.5) through h5py {conda}: filter callback (16 frames after a warm-up chunk), direct-chunk device batches (Python helper and the C entry points), and a dataset of four such batches (on two alternating engine sets); files in memory-backed /dev/shm"
            return out_
        except Exception as e:                                      # (a report, never a gate)
            return {"error": repr(e)}

    def run_batches(data, cfg_, reps=1, keep_streams=()):
        """encode + decode of `data` ((m, H, W) on the device) through ebcc_hip_encode_shard / ebcc_hip_decode_shard: batches of
        n frames on two alternating engine sets (encode: the entropy stage of one batch beside the kernels of the next;
        decode: both sets side by side); returns (encode s, decode s, compressed bytes, frames that keep a residual layer,
        max abs error; run_batches.per_frame holds the max abs error of every frame of the last repetition)."""
        m = data.shape[0]
        dec = torch.empty_like(data)
        te = td = 0.0
        nbytes = resid = 0
        kept = {}
        outs_ = (ctypes.c_void_p * m)()
        sizes_ = (ctypes.c_size_t * m)()
        for _ in range(reps):
            nbytes = resid = 0
            torch.cuda.synchronize()
            t0_ = time.perf_counter()
            assert lib.ebcc_hip_encode_shard(ctx, data.data_ptr(), m, ctypes.byref(cfg_), outs_, sizes_) == 0, lib.ebcc_hip_last_error()
            t1_ = time.perf_counter()
            assert lib.ebcc_hip_decode_shard(ctx, outs_, sizes_, m, dec.data_ptr()) == 0, lib.ebcc_hip_last_error()
            torch.cuda.synchronize()
            t2_ = time.perf_counter()
            te += t1_ - t0_
            td += t2_ - t1_
            for i in range(m):
                nbytes += sizes_[i]
                resid += int.from_bytes(ctypes.string_at(outs_[i] + 16, 8), "little") > 0        # header: coeffs_size
                if i in keep_streams:
                    kept[i] = hashlib.sha256(ctypes.string_at(outs_[i], sizes_[i])).hexdigest()
                lib.free_buffer(outs_[i])
        per_frame = torch.cat([(dec[lo:lo + n] - data[lo:lo + n]).abs().amax(dim=(1, 2)) for lo in range(0, m, n)])
        run_batches.per_frame = per_frame
        run_batches.kept = kept
        return te / reps, td / reps, nbytes, resid, float(per_frame.amax())

    def extra_workloads():
        """Measured in the same run (N = 1 only): the other single-GPU populations of SURVEY section 8(d) and the
        reference's own host-pointer entry points."""
        ex = {}
        # ---- BASELINE configs[2] at its stated size: 4096 frames resident in HBM, RELATIVE_ERROR 1e-3, per-frame amplitude
        #      ramp, batches of the engine's capacity on the two alternating engine sets; the relative bound checked on
        #      EVERY frame, three streams sent to the reference codec with the CPU baseline (byte parity)
        m3 = 4096 if n >= 256 else 4 * n
        data = synth_frames(torch, m3, device, seed=77, ramp=(0.25, 1.75))
        cfg3 = L.make_config((1, H, W), base_cr=BASE_CR, error=1e-3, residual_type=L.RELATIVE_ERROR)
        # (warm-up: one whole pass - the second engine set exists afterwards, and so do the host pages of 2 GB of streams;
        #  the first pass of a process measures the kernel's page zeroing: encode 8.0 - 11.9 GB/s on four leases where every
        #  later pass gives 13.4 - 13.7, tools/gpu/config3_reps.py)
        run_batches(data, cfg3)
        te, td, nb, resid, worst = run_batches(data, cfg3, keep_streams=(1, m3 // 2, m3 - 2))
        rng_ = (data.amax(dim=(1, 2)) - data.amin(dim=(1, 2)))
        rel_ = run_batches.per_frame / rng_
        assert float(rel_.amax()) <= 1e-3 * 1.01 + 1e-6, float(rel_.amax())
        for i, digest in run_batches.kept.items():
            parity_checks.append((f"config3[{i}]", data[i].cpu().numpy(), digest, "rel"))
        ex["config3"] = {"workload": f"{m3} frames 721x1440, base_cr=30 RELATIVE_ERROR=1e-3, amplitude ramp 0.25..1.75, batches of {n}, second pass over the data",
                         "value": round(m3 * FRAME_BYTES / (te + td) / 1e9, 4), "unit": "GB/s",
                         "encode_GBps": round(m3 * FRAME_BYTES / te / 1e9, 4), "decode_GBps": round(m3 * FRAME_BYTES / td / 1e9, 4),
                         "compressed_bytes_per_frame": int(nb / m3), "frames_with_residual_layer": round(resid / m3, 4),
                         "max_error_over_range": round(float(rel_.amax()), 6),                        # per frame: its error / its range
                         "frames_within_bound": int((rel_ <= 1e-3 * 1.01 + 1e-6).sum()), "seconds": round(te + td, 3)}
        del data
        # ---- the population that keeps the residual layer (slope 1.0, amp 0.7)
        mr = min(n, 128)
        data = synth_frames(torch, mr, device, seed=99, slope=1.0, amp=0.7)
        run_batches(data[:min(mr, 32)], cfg)
        te, td, nb, resid, worst = run_batches(data, cfg)
        ex["residual_population"] = {"workload": f"{mr} frames 721x1440 (spectrum slope 1.0, amplitude 0.7), base_cr=30 MAX_ERROR=0.5",
                                     "value": round(mr * FRAME_BYTES / (te + td) / 1e9, 4), "unit": "GB/s",
                                     "encode_GBps": round(mr * FRAME_BYTES / te / 1e9, 4), "decode_GBps": round(mr * FRAME_BYTES / td / 1e9, 4),
                                     "compressed_bytes_per_frame": int(nb / mr), "frames_with_residual_layer": round(resid / mr, 4),
                                     "max_abs_error": round(worst, 5)}
        del data
        # ---- one GPU's share of BASELINE configs[3] (32768 frames over 8 GPUs): 4096 frames resident in HBM, 16 batches of the
        #      engine's capacity, encode + decode, the error bound checked on EVERY frame; four of its frames go to the
        #      reference codec with the CPU baseline below (byte parity of their streams)
        m4 = 4096 if n >= 256 else 4 * n
        data = synth_frames(torch, m4, device, seed=4096)
        te, td, nb, resid, worst = run_batches(data, cfg, keep_streams=(0, m4 // 3, 2 * m4 // 3, m4 - 1))
        assert worst <= MAX_ERR * 1.01 + 1e-3, worst
        ex["shard4096"] = {"workload": f"{m4} frames 721x1440 resident in HBM (one GPU's share of BASELINE configs[3]), base_cr=30 MAX_ERROR=0.5, batches of {n}",
                           "value": round(m4 * FRAME_BYTES / (te + td) / 1e9, 4), "unit": "GB/s",
                           "encode_GBps": round(m4 * FRAME_BYTES / te / 1e9, 4), "decode_GBps": round(m4 * FRAME_BYTES / td / 1e9, 4),
                           "compressed_bytes_per_frame": int(nb / m4), "frames_with_residual_layer": round(resid / m4, 4),
                           "max_abs_error_over_all_frames": round(worst, 5), "frames_within_bound": int((run_batches.per_frame <= MAX_ERR * 1.01 + 1e-3).sum()),
                           "seconds": round(te + td, 3)}
        for i, digest in run_batches.kept.items():
            parity_checks.append((f"shard4096[{i}]", data[i].cpu().numpy(), digest, "max"))
        del data
        # ---- BASELINE configs[4]'s path: EBCC-filtered HDF5 datasets of one frame per chunk, written and read (i) through the
        #      plain filter-308 callback (one chunk per call) and (ii) as device batches of pre-filtered chunks
        #      (ebcc_amd/h5_batch.py: H5Dwrite_chunk / H5Dread_chunk), with the image's conda h5py in a child process
        torch.cuda.empty_cache()                                                  # (the child makes its own engines: 125 GB for two sets)
        lib.ebcc_hip_release_second_set.argtypes = [ctypes.c_void_p]
        lib.ebcc_hip_release_second_set(ctx)                                      # (and this process has no use for its second set any more)
        ex["h5_path"] = h5_path_rates(n)
        # ---- the reference's host-pointer API: pageable host array in, EBCK container in host memory out (PCIe inclusive)
        host = frames.cpu().numpy()
        ccfg = L.make_config((n, H, W), (1, H, W), base_cr=BASE_CR, error=MAX_ERR, residual_type=L.MAX_ERROR)
        lib.ebcc_encode_chunking.restype = ctypes.c_size_t
        lib.ebcc_encode_chunking.argtypes = [ctypes.c_void_p, ctypes.POINTER(L.CodecConfig), L.c_void_pp]
        lib.ebcc_decode_chunking.restype = ctypes.c_size_t
        lib.ebcc_decode_chunking.argtypes = [ctypes.c_void_p, ctypes.c_size_t, L.c_void_pp]
        best = None
        for rep in range(2):                                                      # (the first pass creates the API's own engines)
            o = ctypes.c_void_p()
            t0_ = time.perf_counter()
            nb = lib.ebcc_encode_chunking(host.ctypes.data, ctypes.byref(ccfg), ctypes.byref(o))
            t1_ = time.perf_counter()
            assert nb > 0
            d = ctypes.c_void_p()
            m = lib.ebcc_decode_chunking(o, nb, ctypes.byref(d))
            t2_ = time.perf_counter()
            assert m == n * H * W
            back = np.ctypeslib.as_array(ctypes.cast(d, ctypes.POINTER(ctypes.c_float)), shape=(n, H, W))
            err = float(np.abs(back[::17] - host[::17]).max())
            lib.free_buffer(o)
            lib.free_buffer(d)
            best = (t1_ - t0_, t2_ - t1_, nb, err)
        te, td, nb, err = best
        ex["host_api"] = {"workload": f"ebcc_encode_chunking / ebcc_decode_chunking on a pageable host array of {n}x721x1440 fp32, one frame per chunk",
                          "encode_GBps": round(n * FRAME_BYTES / te / 1e9, 4), "decode_GBps": round(n * FRAME_BYTES / td / 1e9, 4),
                          "round_trip_GBps": round(n * FRAME_BYTES / (te + td) / 1e9, 4), "container_bytes": int(nb), "max_abs_error": round(err, 5)}
        return ex

    def rehearsal_8_ranks():
        """What ONE rank sees when eight share this box's CPU quota (the driver's 8-GPU scaling run cannot be rehearsed on a
        one-GPU box; its host side can): the timed step again in child processes (i) with LOCAL_WORLD_SIZE=8 - the library then
        sizes its host pool from an eighth of the CPUs - and (ii) confined to an eighth of the quota's CPUs (taskset), i.e.
        with an eighth of the CPU TIME as well, which is what eight busy ranks leave each other."""
        import subprocess
        quota = hosts[0]["quota_cpus"] or len(os.sched_getaffinity(0))
        share = max(1, int(quota // 8))
        cpus = sorted(os.sched_getaffinity(0))[:share]
        base = [sys.executable, os.path.abspath(__file__), "--steps", "5", "--warmup", "2", "--no-cpu-baseline", "--no-extras"]
        out_ = {"cpus_per_rank": share, "of_quota": quota}
        for tag, cmd, env in (("pool_sized_for_8_ranks", base, dict(os.environ, LOCAL_WORLD_SIZE="8")),
                              ("confined_to_an_eighth_of_the_cpus", ["taskset", "-c", ",".join(map(str, cpus))] + base, dict(os.environ))):
            try:
                r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
                d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
                h0 = d["host"]["ranks"][0]
                out_[tag] = {"ms_per_step": d["ms_per_step"], "value": d["value"], "encode_GBps": d["encode_GBps"], "decode_GBps": d["decode_GBps"],
                             "pool_threads": h0["pool_threads"], "zstd_core_s_per_step": h0["zstd_core_s_per_step"],
                             "zstd_wait_ms_per_step": h0["zstd_wait_ms_per_step"],
                             "throttled_ms_per_step": (h0.get("cgroup") or {}).get("throttled_ms_per_step")}
            except Exception as e:                                  # (a report, never a gate)
                out_[tag] = {"error": repr(e)}
        return out_

    extras = None
    rehearsal = None
    if rank == 0 and world == 1 and not args.no_extras:
        extras = extra_workloads()
        torch.cuda.empty_cache()
        rehearsal = rehearsal_8_ranks()

    if rank == 0:
        ms_step = elapsed / args.steps * 1e3
        total_frames = n * world
        value = total_frames * FRAME_BYTES / (elapsed / args.steps) / 1e9
        # dominant kernel by total time: tier-1 coding of every code-block (HIP events recorded on the launching
        # stream of each slice engine around its kernels)
        tms, launches = ctypes.c_double(), ctypes.c_long()
        lib.ebcc_hip_timing_read(ctx, b"t1_encode", ctypes.byref(tms), ctypes.byref(launches))
        kern = {}
        for name in (b"t1_encode", b"t1_symbols", b"t1_mq", b"t1_probe_decode", b"rate_alloc", b"j2k_dwt_fwd", b"spiht_encode", b"t1_decode", b"spiht_decode"):
            a, c = ctypes.c_double(), ctypes.c_long()
            lib.ebcc_hip_timing_read(ctx, name, ctypes.byref(a), ctypes.byref(c))
            if c.value:
                kern[name.decode()] = {"ms_avg": round(a.value / c.value, 4), "launches": c.value}

        def kernel_sources_sha():
            import hashlib
            hsh = hashlib.sha256()
            for f in ("j2k_analysis.hip", "t1_core.hpp", "j2k_rate.hip", "residual_dwt.hip", "residual_spiht.hip"):
                hsh.update(open(os.path.join(ROOT, "ebcc_amd", "csrc", f), "rb").read())
            return hsh.hexdigest()[:16]

        def pmc_traffic(frames_per_launch):
            """HBM bytes per launch of the tier-1 encoder from the committed PMC passes (profiles/r04_pmc_tier1.json,
            tools/gpu/profile.sh: separate FETCH_SIZE and WRITE_SIZE runs of `--frames 64` on one slice = 64 frames per
            dispatch; gfx950 correction of the micro-architecture guide: FETCH_SIZE x 2; units of 1 KB).  The file names
            the kernel sources it was taken with; a file that predates the last change to them gives no figure."""
            try:
                pj = json.load(open(os.path.join(ROOT, "profiles", "r04_pmc_tier1.json")))
                if pj.get("kernel_sources_sha") != kernel_sources_sha():
                    return None, "profiles/r04_pmc_tier1.json predates the current tier-1 kernel sources"
                per = float(pj["frames_per_dispatch"])
                per_frame = sum(2.0 * pj["fetch_kb"][k] + pj["write_kb"][k] for k in pj["kernels"]) * 1024.0 / per
                return int(per_frame * frames_per_launch), "profiles/r04_pmc_tier1.json@" + pj["kernel_sources_sha"]
            except Exception as e:
                return None, "no usable PMC file: " + repr(e)

        roof = None
        if launches.value:
            avg_s = tms.value / launches.value / 1e3
            # fp32 read + compressed bytes written per launch (a batch runs as several concurrent slices,
            # each with its own launch)
            algo = (n * FRAME_BYTES + comp) * args.steps / launches.value
            ach = algo / avg_s / 1e9
            traffic, tsrc = pmc_traffic(n * args.steps / launches.value)
            roof = {"bound": "hbm", "kernel": "tier-1 encoder (k_t1_scan + k_t1_rowoffs + k_t1_emit + k_t1_mqrows)", "achieved": round(ach, 3), "peak": 8000.0, "unit": "GB/s",
                    "frac": round(ach / 8000.0, 6), "traffic": traffic, "traffic_source": tsrc, "avg_launch_ms": round(avg_s * 1e3, 4),
                    "algorithmic_bytes_per_launch": algo, "frames_per_launch": n * args.steps / launches.value}
        if roof is not None:
            # HBM bytes of a whole frame round trip (every k_* kernel of encode + decode, counters as above) against the 8.49 MB
            # a round trip has to move: what the search probes cost
            try:
                st = json.load(open(os.path.join(ROOT, "profiles", "r04_step_traffic.json")))
                roof["step_traffic"] = {"bytes_per_frame_round_trip": st["bytes_per_frame_round_trip"],
                                        "algorithmic_bytes_per_frame_round_trip": st["algorithmic_bytes_per_frame_round_trip"],
                                        "source": "profiles/r04_step_traffic.json@" + st.get("kernel_sources_sha", "?"),
                                        "current_sources": st.get("kernel_sources_sha") == kernel_sources_sha()}
            except Exception as e:
                roof["step_traffic"] = None
        # the decode side's dominant kernel: the tier-1 decoder (one launch per batch): compressed bytes in, int32 samples out
        roof_dec = None
        a, c = ctypes.c_double(), ctypes.c_long()
        lib.ebcc_hip_timing_read(ctx, b"t1_decode", ctypes.byref(a), ctypes.byref(c))
        if c.value:
            avg_s = a.value / c.value / 1e3
            algo = (n * FRAME_BYTES + comp) * args.steps / c.value
            # counter traffic of the kernel per launch: its row of the per-kernel step traffic (FETCH_SIZE x 2 + WRITE_SIZE over one
            # warm step of 256 frames, one launch per step: tools/gpu/hbm_table.sh), scaled to the frames of a launch
            dec_traffic, dec_src = None, None
            try:
                st = json.load(open(os.path.join(ROOT, "profiles", "r04_step_traffic.json")))
                if st.get("kernel_sources_sha") == kernel_sources_sha() and "k_t1_decode_lds" in st["by_kernel"]:
                    dec_traffic = int(st["by_kernel"]["k_t1_decode_lds"] * n * args.steps / c.value)
                    dec_src = "profiles/r04_step_traffic.json@" + st["kernel_sources_sha"]
                else:
                    dec_src = "profiles/r04_step_traffic.json predates the current kernel sources"
            except Exception as e:
                dec_src = "no usable PMC file: " + repr(e)
            roof_dec = {"bound": "hbm", "kernel": "tier-1 decoder (k_t1_decode_lds)", "achieved": round(algo / avg_s / 1e9, 3), "peak": 8000.0, "unit": "GB/s",
                        "frac": round(algo / avg_s / 1e9 / 8000.0, 6), "traffic": dec_traffic, "traffic_source": dec_src, "avg_launch_ms": round(avg_s * 1e3, 4),
                        "algorithmic_bytes_per_launch": algo, "frames_per_launch": n * args.steps / c.value,
                        "note": "a serial entropy decoder: latency- and issue-bound by nature; decode_GBps is the phase as a whole"}
        slices = default_slices
        lib.ebcc_hip_host_threads.restype = ctypes.c_int
        line = {
            "metric": "fp32 GB/s encode+decode, 721x1440 ERA5 frames MAX_ERROR=0.5",
            "value": round(value, 4), "unit": "GB/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_step, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{n}-frame batch 721x1440 fp32 per GPU, base_cr=30 MAX_ERROR=0.5 (BASELINE configs[1])",
                       "frames_per_gpu": n, "parallelism": f"frames sharded over {world} GPU(s), no collective",
                       "hw_queues": int(os.environ.get("GPU_MAX_HW_QUEUES", "4")),
                       "encode_slices": slices, "host_pool_threads": lib.ebcc_hip_host_threads(slices),
                       "host_cpus_affinity": len(os.sched_getaffinity(0))},
            "encode_GBps": round(total_frames * FRAME_BYTES * args.steps / enc_t / 1e9, 4),
            "decode_GBps": round(total_frames * FRAME_BYTES * args.steps / dec_t / 1e9, 4),
            "compressed_bytes_per_frame": int(comp / n), "max_abs_error": round(max_err, 5),
            "kernels": kern, "roofline": roof, "roofline_decode": roof_dec,
        }
        # the host side, per rank; and the step time the host alone would allow: every rank's zstd core-seconds over the CPUs
        # the ranks share (the container's quota if there is one) - a run whose ms_per_step sits on it is bound by the host
        cpus_shared = hosts[0]["quota_cpus"] or len(os.sched_getaffinity(0))
        line["host"] = {"ranks": hosts,
                        "zstd_core_s_per_step_all_ranks": round(sum(h["zstd_core_s_per_step"] for h in hosts), 4),
                        "cpus_shared_by_ranks": cpus_shared,
                        "projected_host_bound_ms_per_step": round(sum(h["zstd_core_s_per_step"] for h in hosts) / cpus_shared * 1e3, 2)}
        if rehearsal:
            line["host"]["rehearsal_8_ranks"] = rehearsal
        try:                                                    # HBM-bound kernels, measured alone (tools/gpu/hbm_table.sh): best and worst of the table
            hk = json.load(open(os.path.join(ROOT, "profiles", "r04_hbm_kernels.json")))
            # rows that are HBM measurements: at least 1 MB per frame to move by role, counter traffic within 2x of it either
            # way (below: the input was still in the Infinity Cache; above: re-reads), and not latency-bound by design
            rows = [r for r in hk["kernels"] if r.get("frac_of_6290") is not None and r["algorithmic_bytes"] >= hk["frames_per_dispatch"] * (1 << 20)
                    and 0.5 <= r["counter_over_algorithmic"] <= 2.5 and "latency-bound" not in r.get("what", "")]
            if rows:
                line["hbm_kernels"] = {"source": "profiles/r04_hbm_kernels.json@" + hk.get("kernel_sources_sha", "?"),
                                       "best": max(rows, key=lambda r: r["frac_of_6290"]), "worst": min(rows, key=lambda r: r["frac_of_6290"])}
        except Exception:
            pass
        if extras:
            line.update(extras)
        if not args.no_cpu_baseline and world == 1:               # (rank 0 at N = 1 only: the other ranks would wait at the end)
            try:
                cores = min(16, len(os.sched_getaffinity(0)))
                host_frames = frames[:4].cpu().numpy()
                checks = [(f"configs[1] frame {i}", host_frames[i], h_, "max") for i, h_ in enumerate(getattr(step, "first_hashes", []))] + parity_checks
                line["cpu_baseline"] = cpu_baseline(host_frames, cores, checks)
            except Exception as e:                              # the baseline is a report, never a gate
                line["cpu_baseline"] = {"error": repr(e)}
        print(json.dumps(line), flush=True)
    lib.ebcc_hip_destroy(ctx)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
