"""ctypes bindings shared by the tests.

`oracle()`  -> oracle/libebcc_oracle.so  (CPU restatement; TEST INFRASTRUCTURE, never the product)
`product()` -> ebcc_amd/libh5z_ebcc.so    (the MI355X library under test, called through its C-ABI)
"""
import ctypes
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_SO = os.path.join(ROOT, "oracle", "libebcc_oracle.so")
PRODUCT_SO = os.path.join(ROOT, "ebcc_amd", "libh5z_ebcc.so")
REF_SO = os.path.join(ROOT, "oracle", "_ref", "libh5z_ebcc_ref.so")
REF_SPIHT_SO = os.path.join(ROOT, "oracle", "_ref", "libspiht_ref.so")
GOLDEN = os.path.join(ROOT, "tests", "golden")

c_size_p = ctypes.POINTER(ctypes.c_size_t)
c_void_pp = ctypes.POINTER(ctypes.c_void_p)


class CodecConfig(ctypes.Structure):
    """codec_config_t, /root/reference/src/ebcc_codec.h:32-39 (and ebcc/zarr_filter.py:9-17)."""
    _fields_ = [
        ("dims", ctypes.c_size_t * 3),
        ("base_cr", ctypes.c_float),
        ("residual_compression_type", ctypes.c_int),
        ("residual_cr", ctypes.c_float),
        ("error", ctypes.c_float),
        ("chunk_dims", ctypes.c_size_t * 3),
    ]


NONE, MAX_ERROR, RELATIVE_ERROR = 0, 1, 2


def make_config(shape, chunk_shape=None, *, base_cr=2.0, error=0.01, residual_type=MAX_ERROR):
    c = CodecConfig()
    c.dims[:] = shape
    c.base_cr = base_cr
    c.residual_compression_type = residual_type
    c.residual_cr = 0.0
    c.error = error
    c.chunk_dims[:] = chunk_shape or (0, 0, 0)
    return c


_oracle = None


def oracle():
    global _oracle
    if _oracle is None:
        if not os.path.exists(ORACLE_SO):
            subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "oracle"])
        lib = ctypes.CDLL(ORACLE_SO)
        lib.orc_spiht_encode.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_size_t, c_void_pp, c_size_p,
                                         ctypes.c_size_t, ctypes.c_size_t]
        lib.orc_spiht_decode.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_size_t,
                                         ctypes.c_size_t, ctypes.c_size_t]
        lib.orc_spiht_analysis.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_size_t, ctypes.c_size_t,
                                           ctypes.c_void_p]
        lib.orc_spiht_analysis.restype = ctypes.c_int
        for name in ("orc_ebcc_encode", "orc_ebcc_encode_chunking", "orc_ebcc_encode_chunking_compat"):
            f = getattr(lib, name)
            f.restype = ctypes.c_size_t
            f.argtypes = [ctypes.c_void_p, ctypes.POINTER(CodecConfig), c_void_pp]
        for name in ("orc_ebcc_decode", "orc_ebcc_decode_chunking"):
            f = getattr(lib, name)
            f.restype = ctypes.c_size_t
            f.argtypes = [ctypes.c_void_p, ctypes.c_size_t, c_void_pp]
        lib.orc_free.argtypes = [ctypes.c_void_p]
        lib.orc_j2k_encode.restype = ctypes.c_size_t
        lib.orc_j2k_encode.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_size_t, ctypes.c_float, c_void_pp]
        lib.orc_j2k_decode.restype = ctypes.c_size_t
        lib.orc_j2k_decode.argtypes = [ctypes.c_void_p, ctypes.c_size_t, c_void_pp, c_size_p, c_size_p]
        if hasattr(lib, "orc_opj_encode"):
            lib.orc_opj_encode.restype = ctypes.c_size_t
            lib.orc_opj_encode.argtypes = lib.orc_j2k_encode.argtypes
            lib.orc_opj_decode.restype = ctypes.c_size_t
            lib.orc_opj_decode.argtypes = lib.orc_j2k_decode.argtypes
        _oracle = lib
    return _oracle


def padded_shape(h, w, stages=3):
    u = 1 << (stages + 1)
    return (h + (-h) % u, w + (-w) % u)


def orc_spiht_encode(img, trunc_bits, stages=3):
    lib = oracle()
    img = np.ascontiguousarray(img, np.float32)
    h, w = img.shape
    buf, n = ctypes.c_void_p(), ctypes.c_size_t()
    lib.orc_spiht_encode(img.ctypes.data, h, w, ctypes.byref(buf), ctypes.byref(n), trunc_bits, stages)
    s = ctypes.string_at(buf.value, n.value)
    lib.orc_free(buf)
    return s


def orc_spiht_decode(stream, h, w, num_bits=None):
    lib = oracle()
    out = np.zeros((h, w), np.float32)
    b = ctypes.create_string_buffer(bytes(stream), len(stream))
    lib.orc_spiht_decode(b, len(stream), out.ctypes.data, h, w, 8 * len(stream) if num_bits is None else num_bits)
    return out


def orc_spiht_coeffs(img, stages=3):
    lib = oracle()
    img = np.ascontiguousarray(img, np.float32)
    h, w = img.shape
    ph, pw = padded_shape(h, w, stages)
    c = np.zeros((ph, pw), np.float32)
    dc = lib.orc_spiht_analysis(img.ctypes.data, h, w, stages, c.ctypes.data)
    return c.astype(np.int32), dc


def _take(lib, ptr, n, free):
    s = ctypes.string_at(ptr.value, n)
    free(ptr)
    return s


def orc_encode(data, cfg, fn="orc_ebcc_encode"):
    lib = oracle()
    data = np.ascontiguousarray(data, np.float32)
    out = ctypes.c_void_p()
    n = getattr(lib, fn)(data.ctypes.data, ctypes.byref(cfg), ctypes.byref(out))
    if n == 0:
        return b""
    return _take(lib, out, n, lib.orc_free)


def orc_decode(stream, fn="orc_ebcc_decode"):
    lib = oracle()
    b = ctypes.create_string_buffer(bytes(stream), len(stream))
    out = ctypes.c_void_p()
    n = getattr(lib, fn)(b, len(stream), ctypes.byref(out))
    if n == 0:
        return None
    a = np.frombuffer(ctypes.string_at(out.value, 4 * n), np.float32).copy()
    lib.orc_free(out)
    return a


# ------------------------------------------------------------------------------------------ product
_product = None


def product():
    """The MI355X library.  No fallback: a missing/unloadable library is an error."""
    global _product
    if _product is None:
        lib = ctypes.CDLL(PRODUCT_SO)
        lib.ebcc_hip_create.restype = ctypes.c_void_p
        lib.ebcc_hip_create.argtypes = [ctypes.c_int, ctypes.c_size_t, ctypes.c_size_t, ctypes.c_size_t]
        lib.ebcc_hip_destroy.argtypes = [ctypes.c_void_p]
        lib.ebcc_hip_malloc.restype = ctypes.c_void_p
        lib.ebcc_hip_malloc.argtypes = [ctypes.c_size_t]
        lib.ebcc_hip_free.argtypes = [ctypes.c_void_p]
        lib.ebcc_hip_memcpy_h2d.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t]
        lib.ebcc_hip_memcpy_d2h.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t]
        lib.ebcc_hip_padded_pixels.restype = ctypes.c_size_t
        lib.ebcc_hip_padded_pixels.argtypes = [ctypes.c_void_p]
        lib.ebcc_hip_workspace_bytes.restype = ctypes.c_size_t
        lib.ebcc_hip_workspace_bytes.argtypes = [ctypes.c_void_p]
        lib.ebcc_hip_spiht_encode.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, c_size_p, c_void_pp,
                                              c_size_p]
        lib.ebcc_hip_spiht_decode.argtypes = [ctypes.c_void_p, c_void_pp, c_size_p, c_size_p, ctypes.c_size_t,
                                              ctypes.c_void_p]
        lib.ebcc_hip_spiht_decode_prefix.argtypes = [ctypes.c_void_p, ctypes.c_size_t, c_size_p, ctypes.c_void_p]
        lib.ebcc_hip_spiht_coeffs.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p,
                                              ctypes.c_void_p]
        lib.ebcc_hip_last_error.restype = ctypes.c_char_p
        if hasattr(lib, "ebcc_hip_j2k_encode"):
            lib.ebcc_hip_j2k_encode.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p, c_void_pp,
                                                c_size_p, ctypes.c_void_p]
            lib.ebcc_hip_j2k_emulated_decode.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p,
                                                         ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
            lib.ebcc_hip_j2k_decode.argtypes = [ctypes.c_void_p, c_void_pp, c_size_p, ctypes.c_size_t, ctypes.c_void_p,
                                                ctypes.c_void_p]
        lib.free_buffer.argtypes = [ctypes.c_void_p]
        if hasattr(lib, "ebcc_hip_encode_frames"):
            lib.ebcc_hip_encode_frames.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t,
                                                   ctypes.POINTER(CodecConfig), c_void_pp, c_size_p]
            lib.ebcc_hip_decode_frames.argtypes = [ctypes.c_void_p, c_void_pp, c_size_p, ctypes.c_size_t,
                                                   ctypes.c_void_p]
        if hasattr(lib, "ebcc_hip_encode_shard"):
            lib.ebcc_hip_encode_shard.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t,
                                                  ctypes.POINTER(CodecConfig), c_void_pp, c_size_p]
            lib.ebcc_hip_decode_shard.argtypes = [ctypes.c_void_p, c_void_pp, c_size_p, ctypes.c_size_t, ctypes.c_void_p]
        for name in ("ebcc_encode", "ebcc_encode_chunking", "ebcc_encode_chunking_compat"):
            if hasattr(lib, name):
                f = getattr(lib, name)
                f.restype = ctypes.c_size_t
                f.argtypes = [ctypes.c_void_p, ctypes.POINTER(CodecConfig), c_void_pp]
        for name in ("ebcc_decode", "ebcc_decode_chunking"):
            if hasattr(lib, name):
                f = getattr(lib, name)
                f.restype = ctypes.c_size_t
                f.argtypes = [ctypes.c_void_p, ctypes.c_size_t, c_void_pp]
        _product = lib
    return _product


class DeviceArray:
    """A device buffer owned through the product's own malloc helpers (no torch needed)."""

    def __init__(self, host=None, nbytes=None):
        lib = product()
        self.nbytes = host.nbytes if host is not None else nbytes
        self.ptr = lib.ebcc_hip_malloc(self.nbytes)
        assert self.ptr, lib.ebcc_hip_last_error()
        if host is not None:
            host = np.ascontiguousarray(host)
            assert lib.ebcc_hip_memcpy_h2d(self.ptr, host.ctypes.data, host.nbytes) == 0

    def get(self, dtype, shape):
        out = np.empty(shape, dtype)
        assert out.nbytes <= self.nbytes
        assert product().ebcc_hip_memcpy_d2h(out.ctypes.data, self.ptr, out.nbytes) == 0
        return out

    def free(self):
        if self.ptr:
            product().ebcc_hip_free(self.ptr)
            self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class Context:
    def __init__(self, max_frames, h, w, device=0):
        self.lib = product()
        self.h, self.w, self.max_frames = h, w, max_frames
        self.ptr = self.lib.ebcc_hip_create(device, max_frames, h, w)
        assert self.ptr, self.lib.ebcc_hip_last_error()

    def close(self):
        if self.ptr:
            self.lib.ebcc_hip_destroy(self.ptr)
            self.ptr = None

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    # ---- residual layer
    def spiht_encode(self, images, trunc_bits):
        images = np.ascontiguousarray(images, np.float32)
        n = images.shape[0]
        d = DeviceArray(images)
        tb = (ctypes.c_size_t * n)(*trunc_bits)
        outs = (ctypes.c_void_p * n)()
        sizes = (ctypes.c_size_t * n)()
        rc = self.lib.ebcc_hip_spiht_encode(self.ptr, d.ptr, n, tb, outs, sizes)
        assert rc == 0, self.lib.ebcc_hip_last_error()
        res = []
        for f in range(n):
            res.append(ctypes.string_at(outs[f], sizes[f]))
            self.lib.free_buffer(outs[f])
        d.free()
        return res

    def spiht_decode(self, streams, num_bits=None):
        n = len(streams)
        bufs = [ctypes.create_string_buffer(bytes(s), len(s)) for s in streams]
        ptrs = (ctypes.c_void_p * n)(*[ctypes.cast(b, ctypes.c_void_p).value for b in bufs])
        sizes = (ctypes.c_size_t * n)(*[len(s) for s in streams])
        nb = (ctypes.c_size_t * n)(*[8 * len(s) for s in streams] if num_bits is None else num_bits)
        out = DeviceArray(nbytes=n * self.h * self.w * 4)
        rc = self.lib.ebcc_hip_spiht_decode(self.ptr, ptrs, sizes, nb, n, out.ptr)
        assert rc == 0, self.lib.ebcc_hip_last_error()
        res = out.get(np.float32, (n, self.h, self.w))
        out.free()
        return res

    def spiht_decode_prefix(self, n, trunc_bits):
        tb = (ctypes.c_size_t * n)(*trunc_bits)
        out = DeviceArray(nbytes=n * self.h * self.w * 4)
        rc = self.lib.ebcc_hip_spiht_decode_prefix(self.ptr, n, tb, out.ptr)
        assert rc == 0, self.lib.ebcc_hip_last_error()
        res = out.get(np.float32, (n, self.h, self.w))
        out.free()
        return res

    def spiht_coeffs(self, images):
        images = np.ascontiguousarray(images, np.float32)
        n = images.shape[0]
        d = DeviceArray(images)
        npad = self.lib.ebcc_hip_padded_pixels(self.ptr)
        c = np.zeros((n, npad), np.int32)
        dc = np.zeros(n, np.int32)
        rc = self.lib.ebcc_hip_spiht_coeffs(self.ptr, d.ptr, n, c.ctypes.data, dc.ctypes.data)
        assert rc == 0, self.lib.ebcc_hip_last_error()
        d.free()
        return c, dc

    # ---- JPEG 2000 base layer
    def j2k_encode(self, frames, cr, keep_device=False):
        frames = np.ascontiguousarray(frames, np.float32)
        n = frames.shape[0]
        d = DeviceArray(frames)
        crs = np.ascontiguousarray(cr, np.float32)
        outs = (ctypes.c_void_p * n)()
        sizes = (ctypes.c_size_t * n)()
        mm = np.zeros(2 * n, np.float32)
        rc = self.lib.ebcc_hip_j2k_encode(self.ptr, d.ptr, n, crs.ctypes.data, outs, sizes, mm.ctypes.data)
        assert rc == 0, self.lib.ebcc_hip_last_error()
        res = []
        for f in range(n):
            res.append(ctypes.string_at(outs[f], sizes[f]))
            self.lib.free_buffer(outs[f])
        if keep_device:
            return res, mm.reshape(n, 2), d
        d.free()
        return res, mm.reshape(n, 2)

    def j2k_emulated_decode(self, d_frames, n, target):
        tg = np.ascontiguousarray(target, np.float32)
        out = DeviceArray(nbytes=n * self.h * self.w * 4)
        nbad = np.zeros(n, np.uint64)
        esum = np.zeros(n, np.float64)
        rc = self.lib.ebcc_hip_j2k_emulated_decode(self.ptr, d_frames.ptr, n, tg.ctypes.data, out.ptr, nbad.ctypes.data,
                                                   esum.ctypes.data)
        assert rc == 0, self.lib.ebcc_hip_last_error()
        res = out.get(np.float32, (n, self.h, self.w))
        out.free()
        return res, nbad, esum

    def j2k_decode(self, streams, minmax):
        n = len(streams)
        bufs = [ctypes.create_string_buffer(bytes(s), len(s)) for s in streams]
        ptrs = (ctypes.c_void_p * n)(*[ctypes.cast(b, ctypes.c_void_p).value for b in bufs])
        sizes = (ctypes.c_size_t * n)(*[len(s) for s in streams])
        mm = np.ascontiguousarray(minmax, np.float32)
        out = DeviceArray(nbytes=n * self.h * self.w * 4)
        rc = self.lib.ebcc_hip_j2k_decode(self.ptr, ptrs, sizes, n, mm.ctypes.data, out.ptr)
        assert rc == 0, self.lib.ebcc_hip_last_error()
        res = out.get(np.float32, (n, self.h, self.w))
        out.free()
        return res

    # ---- frame codec
    def encode_frames(self, frames, cfg):
        frames = np.ascontiguousarray(frames, np.float32)
        n = frames.shape[0]
        d = DeviceArray(frames)
        outs = (ctypes.c_void_p * n)()
        sizes = (ctypes.c_size_t * n)()
        rc = self.lib.ebcc_hip_encode_frames(self.ptr, d.ptr, n, ctypes.byref(cfg), outs, sizes)
        assert rc == 0, self.lib.ebcc_hip_last_error()
        res = []
        for f in range(n):
            res.append(ctypes.string_at(outs[f], sizes[f]))
            self.lib.free_buffer(outs[f])
        d.free()
        return res

    def encode_shard(self, frames, cfg):
        """any number of frames through ebcc_hip_encode_shard (batches of the context's capacity on two engine sets)"""
        frames = np.ascontiguousarray(frames, np.float32)
        n = frames.shape[0]
        d = DeviceArray(frames)
        outs = (ctypes.c_void_p * n)()
        sizes = (ctypes.c_size_t * n)()
        rc = self.lib.ebcc_hip_encode_shard(self.ptr, d.ptr, n, ctypes.byref(cfg), outs, sizes)
        d.free()
        if rc:
            assert all(not outs[f] for f in range(n))                  # (freed by the library)
            return None
        res = []
        for f in range(n):
            res.append(ctypes.string_at(outs[f], sizes[f]))
            self.lib.free_buffer(outs[f])
        return res

    def decode_frames(self, streams, shard=False):
        n = len(streams)
        bufs = [ctypes.create_string_buffer(bytes(s), len(s)) for s in streams]
        ptrs = (ctypes.c_void_p * n)(*[ctypes.cast(b, ctypes.c_void_p).value for b in bufs])
        sizes = (ctypes.c_size_t * n)(*[len(s) for s in streams])
        out = DeviceArray(nbytes=n * self.h * self.w * 4)
        rc = (self.lib.ebcc_hip_decode_shard if shard else self.lib.ebcc_hip_decode_frames)(self.ptr, ptrs, sizes, n, out.ptr)
        assert rc == 0, self.lib.ebcc_hip_last_error()
        res = out.get(np.float32, (n, self.h, self.w))
        out.free()
        return res


def legacy_repack(stream):
    """The same frame in the reference's legacy header-less layout (src/ebcc_codec.c:1147-1213):
    f32 min, f32 max, u64 coeffs_size, f32 rmin, f32 rmax, u64 compressed_size, zstd payload, tail."""
    import struct
    magic, ver, flags, _r, mn, mx, coeffs, rmn, rmx, zsize, tsize = struct.unpack("<4sBBHIIQIIQQ", stream[:48])
    assert magic == b"EBCC" and ver == 1 and len(stream) == 48 + zsize + tsize
    return struct.pack("<IIQIIQ", mn, mx, coeffs, rmn, rmx, zsize) + stream[48:]


# ------------------------------------------------------------------------------------------ inputs
def scale_u16(field):
    """ebcc_codec.c:686-689 in float32 arithmetic."""
    field = np.ascontiguousarray(field, np.float32)
    mn, mx = field.min(), field.max()
    return (((field - mn) / (mx - mn)) * np.float32(65535)).astype(np.uint16), mn, mx


def opj_version():
    """Version string of the libopenjp2 the oracle's optional backend found ('' = none)."""
    lib = oracle()
    lib.orc_opj_version.restype = ctypes.c_char_p
    return (lib.orc_opj_version() or b"").decode()


def orc_j2k_encode(img_u16, cr):
    lib = oracle()
    img = np.ascontiguousarray(img_u16, np.uint16)
    out = ctypes.c_void_p()
    n = lib.orc_j2k_encode(img.ctypes.data, img.shape[0], img.shape[1], ctypes.c_float(cr), ctypes.byref(out))
    s = ctypes.string_at(out.value, n)
    lib.orc_free(out)
    return s


def orc_j2k_decode(stream):
    lib = oracle()
    b = ctypes.create_string_buffer(bytes(stream), len(stream))
    out, h, w = ctypes.c_void_p(), ctypes.c_size_t(), ctypes.c_size_t()
    n = lib.orc_j2k_decode(b, len(stream), ctypes.byref(out), ctypes.byref(h), ctypes.byref(w))
    assert n
    a = np.frombuffer(ctypes.string_at(out.value, 4 * n), np.int32).reshape(h.value, w.value).copy()
    lib.orc_free(out)
    return a


def map_decoded(samples, mn, mx):
    """ebcc_codec.c:1130 in float32 arithmetic."""
    return (samples.astype(np.float32) / np.float32(65535)) * (np.float32(mx) - np.float32(mn)) + np.float32(mn)


def kat_image(h, w):
    """a[y,x] = float32((7x+13y) mod 97) / float32(96)."""
    y, x = np.mgrid[0:h, 0:w]
    return (((7 * x + 13 * y) % 97).astype(np.float32) / np.float32(96)).astype(np.float32)


def smooth_image(h, w, seed):
    r = np.random.default_rng(seed)
    a = np.cumsum(np.cumsum(r.standard_normal((h, w)), axis=1), axis=0)
    a = (a - a.min()) / (a.max() - a.min())
    return a.astype(np.float32)


def era5_like(h, w, seed, slope=1.5, amp=2.5):
    """SURVEY.md section 8(d) synthetic generator (k^-slope spectrum on a zonal profile)."""
    r = np.random.default_rng(seed)
    wn = r.standard_normal((h, w))
    f = np.fft.rfft2(wn)
    ky = np.fft.fftfreq(h)[:, None]
    kx = np.fft.rfftfreq(w)[None, :]
    k = np.sqrt(ky * ky + kx * kx)
    k[0, 0] = 1
    f *= k ** (-slope)
    f[0, 0] = 0
    n = np.fft.irfft2(f, s=(h, w))
    n /= n.std()
    lat = np.linspace(-1, 1, h)[:, None]
    return (235 + 50 * np.cos(lat * np.pi / 2) + amp * n).astype(np.float32)


def formula_frames(n, h, w):
    """Integer-formula frames (exactly reproducible anywhere): frame k of an n x h x w stack."""
    k, y, x = np.mgrid[0:n, 0:h, 0:w]
    return (250.0 + ((x * 3 + y * 5 + k * 11) % 1024).astype(np.float32) / np.float32(64.0)
            + (((x // 16) * 7 + (y // 16) * 13 + k * 5) % 97).astype(np.float32)).astype(np.float32)


def _orc_encode_job(arg):
    frame, cfg_fields = arg
    oracle().orc_set_j2k_backend(0)
    cfg = make_config(cfg_fields[0], base_cr=cfg_fields[1], error=cfg_fields[2], residual_type=cfg_fields[3])
    return orc_encode(frame, cfg)


def orc_encode_many(frames, cfg, workers=6):
    """orc_encode of several single-frame inputs side by side (the restated JPEG 2000 search takes seconds per full-size
    frame).  Spawned workers: a process that has initialised HIP is never forked; the workers load the oracle only."""
    import multiprocessing as mp
    fields = (tuple(cfg.dims), float(cfg.base_cr), float(cfg.error), int(cfg.residual_compression_type))
    jobs = [(np.ascontiguousarray(f, np.float32), fields) for f in frames]
    with mp.get_context("spawn").Pool(min(workers, len(jobs))) as pool:
        return pool.map(_orc_encode_job, jobs, chunksize=1)
