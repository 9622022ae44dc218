"""HDF5 direct-chunk batch path.

HDF5 calls a filter once per chunk (`H5Z_filter_ebcc`, /root/reference/src/h5z_ebcc.c:124-148): one frame per
call, which leaves a GPU codec launch-latency-bound.  For datasets whose chunks are single frames (1, H, W) the
chunks can instead be coded as one batch on the device and handed to HDF5 already filtered
(`H5Dwrite_chunk` / `H5Dread_chunk`, h5py's `write_direct_chunk` / `read_direct_chunk`).  The file is an ordinary
EBCC-filtered dataset: the bytes of every chunk are exactly what the filter callback would have produced, so
any HDF5 reader with the plugin on `HDF5_PLUGIN_PATH` (this build or the reference's) reads it, and files
written through the callback are read here.

Only numpy and ctypes are needed (h5py objects are passed in by the caller).
"""
import ctypes
import os
import sys
import threading
import time

import numpy as np

from . import load
from .filter_wrapper import EBCC_Filter, _MODES


class CodecConfig(ctypes.Structure):
    """codec_config_t, include/ebcc_codec.h (reference src/ebcc_codec.h:32-39)."""
    _fields_ = [("dims", ctypes.c_size_t * 3), ("base_cr", ctypes.c_float), ("residual_compression_type", ctypes.c_int),
                ("residual_cr", ctypes.c_float), ("error", ctypes.c_float), ("chunk_dims", ctypes.c_size_t * 3)]


def frame_config(height, width, base_cr, residual_opt=("none", None)):
    mode, value = residual_opt if residual_opt is not None else ("none", None)
    c = CodecConfig()
    c.dims[:] = (1, height, width)
    c.base_cr = base_cr
    c.residual_compression_type = _MODES[mode]
    c.residual_cr = 0.0
    c.error = float(value) if _MODES[mode] else 0.0
    c.chunk_dims[:] = (0, 0, 0)
    return c


class BatchCodec:
    """Device engine for stacks of (H, W) float32 frames held in host memory."""

    def __init__(self, height, width, max_frames=256, device=0):
        lib = self.lib = load()
        lib.ebcc_hip_create.restype = ctypes.c_void_p
        lib.ebcc_hip_create.argtypes = [ctypes.c_int, ctypes.c_size_t, ctypes.c_size_t, ctypes.c_size_t]
        lib.ebcc_hip_destroy.argtypes = [ctypes.c_void_p]
        lib.ebcc_hip_malloc.restype = ctypes.c_void_p
        lib.ebcc_hip_malloc.argtypes = [ctypes.c_size_t]
        lib.ebcc_hip_free.argtypes = [ctypes.c_void_p]
        lib.ebcc_hip_memcpy_h2d.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t]
        lib.ebcc_hip_memcpy_d2h.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t]
        lib.ebcc_hip_encode_frames.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.POINTER(CodecConfig),
                                               ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_size_t)]
        lib.ebcc_hip_decode_frames.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_size_t),
                                               ctypes.c_size_t, ctypes.c_void_p]
        lib.ebcc_hip_upload.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t]
        lib.ebcc_hip_download.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t]
        lib.ebcc_hip_encode_host_frames.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.POINTER(CodecConfig),
                                                    ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_size_t)]
        lib.ebcc_hip_decode_host_frames.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_size_t),
                                                    ctypes.c_size_t, ctypes.c_void_p]
        lib.ebcc_hip_last_error.restype = ctypes.c_char_p
        lib.free_buffer.argtypes = [ctypes.c_void_p]
        self.h, self.w, self.max_frames = int(height), int(width), int(max_frames)
        self.ctx = lib.ebcc_hip_create(device, self.max_frames, self.h, self.w)
        if not self.ctx:
            raise RuntimeError("EBCC MI355X engine: " + (lib.ebcc_hip_last_error() or b"?").decode())

    def close(self):
        if self.ctx:
            self.lib.ebcc_hip_destroy(self.ctx)
            self.ctx = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def encode(self, frames, cfg):
        """frames: (n, H, W) float32 in host memory, any n -> list of EBCC frame streams (bytes).  More frames than the
        engine holds are coded in batches on two alternating engine sets (the host part of one batch beside the upload and
        the kernels of the next)."""
        frames = np.ascontiguousarray(frames, np.float32)
        n = frames.shape[0]
        assert frames.shape[1:] == (self.h, self.w) and n >= 1
        outs = (ctypes.c_void_p * n)()
        sizes = (ctypes.c_size_t * n)()
        if self.lib.ebcc_hip_encode_host_frames(self.ctx, frames.ctypes.data, n, ctypes.byref(cfg), outs, sizes):
            raise RuntimeError("ebcc_hip_encode_host_frames: " + (self.lib.ebcc_hip_last_error() or b"?").decode())
        res = []
        for i in range(n):
            res.append(ctypes.string_at(outs[i], sizes[i]))
            self.lib.free_buffer(outs[i])
        return res

    def decode(self, streams, out=None):
        """list of EBCC frame streams (bytes), any number -> (n, H, W) float32; `out`: a C-contiguous float32 array to decode
        into (the frames cross PCIe straight into it; its pages are mapped while the GPU decodes, one batch is downloaded
        beside the kernels of the next)."""
        n = len(streams)
        assert n >= 1
        streams = [s if isinstance(s, bytes) else bytes(s) for s in streams]
        # (pointers into the bytes objects themselves: they stay alive in `streams` for the duration of the call)
        ptrs = (ctypes.c_void_p * n)(*[ctypes.cast(ctypes.c_char_p(s), ctypes.c_void_p).value for s in streams])
        sizes = (ctypes.c_size_t * n)(*[len(s) for s in streams])
        if out is None:
            out = np.empty((n, self.h, self.w), np.float32)
        assert out.dtype == np.float32 and out.flags.c_contiguous and out.size == n * self.h * self.w
        t0 = time.perf_counter()
        if self.lib.ebcc_hip_decode_host_frames(self.ctx, ptrs, sizes, n, out.ctypes.data):
            raise RuntimeError("ebcc_hip_decode_host_frames: " + (self.lib.ebcc_hip_last_error() or b"?").decode())
        if os.environ.get("EBCC_H5_TIMING"):
            print(f"h5_batch.decode: {n} frames decoded and downloaded in {1e3 * (time.perf_counter() - t0):.1f} ms", file=sys.stderr, flush=True)
        return out


_SUPER = 16         # batches handed to the engine per call by write_frames / read_frames (filling and draining the two sets costs ~half a batch)

# Engines are expensive to make (tens of GB of device workspace for 256 frames of 721 x 1440) and cheap to keep: the
# helpers below share ONE per (height, width, device) for the life of the process - a codec made for more frames serves a
# request for fewer, a larger request replaces it (the old one is closed first), and when the device has no room for a new
# engine every cached one is closed and the creation is tried once more.  close_cached() gives the memory back; callers
# that hold datasets of many geometries in one process and want the memory between them call it themselves.
_codecs = {}


def cached_codec(height, width, max_frames=256, device=0):
    key = (int(height), int(width), int(device))
    c = _codecs.get(key)
    if c is not None and c.ctx and c.max_frames >= int(max_frames):
        return c
    if c is not None:
        c.close()
        del _codecs[key]
    try:
        c = BatchCodec(height, width, max_frames, device)
    except RuntimeError:
        close_cached()                                          # (other geometries' engines may hold the memory)
        c = BatchCodec(height, width, max_frames, device)
    _codecs[key] = c
    return c


def close_cached():
    for c in _codecs.values():
        c.close()
    _codecs.clear()


import atexit  # noqa: E402

atexit.register(close_cached)


def create_dataset(group, name, shape, base_cr, residual_opt=("none", None), **kw):
    """An EBCC-filtered float32 dataset of shape (..., H, W) with one frame per chunk."""
    h, w = shape[-2:]
    return group.create_dataset(name, shape=shape,
                                **EBCC_Filter(base_cr=base_cr, height=h, width=w, residual_opt=residual_opt, data_dim=len(shape)), **kw)


def write_frames(dset, data, base_cr, residual_opt=("none", None), batch=256, codec=None):
    """Code `data` (same shape as `dset`, frames in its last two axes) in device batches and store every frame
    as a pre-filtered chunk.  `base_cr` / `residual_opt` must be the dataset's filter parameters."""
    data = np.asarray(data, np.float32)
    assert tuple(data.shape) == tuple(dset.shape) and dset.chunks == (1,) * (data.ndim - 2) + data.shape[-2:]
    h, w = data.shape[-2:]
    flat = data.reshape((-1, h, w))
    lead = data.shape[:-2]
    cfg = frame_config(h, w, base_cr, residual_opt)
    codec = codec or cached_codec(h, w, min(batch, len(flat)))
    step = _SUPER * codec.max_frames                            # (several batches per call: they alternate between two engine sets)
    for lo in range(0, len(flat), step):
        streams = codec.encode(flat[lo:lo + step], cfg)
        for i, s in enumerate(streams):
            idx = np.unravel_index(lo + i, lead) if lead else ()
            dset.id.write_direct_chunk(tuple(int(v) for v in idx) + (0, 0), s, filter_mask=0)


def read_frames(dset, batch=256, codec=None):
    """Read an EBCC-filtered one-frame-per-chunk dataset by decoding its raw chunks in device batches, several batches per
    call (they alternate between two engine sets: one is downloaded while the next decodes).  The raw chunks of the next
    call are fetched from the file (h5py, one call per chunk) on a helper thread meanwhile; the frames land straight in
    their place in the result."""
    h, w = dset.shape[-2:]
    lead = dset.shape[:-2]
    n = int(np.prod(lead)) if lead else 1
    out = np.empty((n, h, w), np.float32)
    codec = codec or cached_codec(h, w, min(batch, n))
    step = _SUPER * codec.max_frames

    def fetch(lo, box):
        try:
            raw = []
            for i in range(lo, min(n, lo + step)):
                idx = np.unravel_index(i, lead) if lead else ()
                mask, chunk = dset.id.read_direct_chunk(tuple(int(v) for v in idx) + (0, 0))
                if mask:
                    raise ValueError(f"chunk {idx} was stored with filters disabled (mask {mask})")
                raw.append(chunk)
            box.append(raw)
        except BaseException as e:                                  # (handed to the caller's thread)
            box.append(e)

    box = []
    t0 = time.perf_counter()
    fetch(0, box)
    if os.environ.get("EBCC_H5_TIMING"):
        print(f"h5_batch.read_frames: first {min(n, step)} chunks fetched in {1e3 * (time.perf_counter() - t0):.1f} ms", file=sys.stderr, flush=True)
    for lo in range(0, n, step):
        raw = box[0]
        if isinstance(raw, BaseException):
            raise raw
        box, t = [], None
        if lo + step < n:
            t = threading.Thread(target=fetch, args=(lo + step, box))
            t.start()
        try:
            codec.decode(raw, out=out[lo:lo + len(raw)])
        finally:
            if t:
                t.join()
    return out.reshape(dset.shape)
