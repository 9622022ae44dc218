"""Zarr / numcodecs codec on the MI355X library.

Same contract as the reference's `ebcc.zarr_filter.EBCCZarrFilter` (/root/reference/ebcc/zarr_filter.py:19-92):
codec id "ebcc_filter", one constructor argument `arglist` = the HDF5 `cd_values`
(`EBCC_Filter(...)["compression_opts"]`: H, W, f32 bits of base_cr, mode, [f32 bits of error]), `encode` takes a
float32 array holding whole frames and returns bytes, `decode` returns a flat float32 array or fills `out`, and
`get_config` / `from_config` round-trip `{"id", "arglist"}`.  numcodecs is optional: when it is importable the
class derives from `numcodecs.abc.Codec` and registers itself, otherwise it is a plain class with the same
methods (the C library does the work either way).
"""
import ctypes

import numpy as np

from . import load
from .h5_batch import CodecConfig

try:                                                     # pragma: no cover - depends on the environment
    import numcodecs
    from numcodecs.abc import Codec as _Base
except ImportError:                                      # numcodecs is not part of this image
    numcodecs = None
    _Base = object


class EBCCZarrFilter(_Base):
    codec_id = "ebcc_filter"

    def __init__(self, arglist):
        self.arglist = np.asarray(arglist, dtype=np.uint32)
        if self.arglist.ndim != 1 or len(self.arglist) not in (4, 5):
            raise ValueError("arglist must hold the 4 or 5 EBCC filter values (H, W, base_cr bits, mode[, error bits])")
        lib = self._lib = load()
        lib.populate_config.restype = None
        lib.populate_config.argtypes = [ctypes.POINTER(CodecConfig), ctypes.c_size_t, ctypes.c_void_p, ctypes.c_size_t]
        lib.ebcc_encode.restype = ctypes.c_size_t
        lib.ebcc_encode.argtypes = [ctypes.c_void_p, ctypes.POINTER(CodecConfig), ctypes.POINTER(ctypes.c_void_p)]
        lib.ebcc_decode.restype = ctypes.c_size_t
        lib.ebcc_decode.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.POINTER(ctypes.c_void_p)]
        lib.free_buffer.argtypes = [ctypes.c_void_p]

    def encode(self, buf):
        a = np.asarray(buf)
        if a.dtype != np.float32:
            raise TypeError("EBCC codes float32 data")
        a = np.ascontiguousarray(a).reshape(-1)
        cfg = CodecConfig()
        self._lib.populate_config(ctypes.byref(cfg), len(self.arglist), self.arglist.ctypes.data, a.nbytes)
        out = ctypes.c_void_p()
        n = self._lib.ebcc_encode(a.ctypes.data, ctypes.byref(cfg), ctypes.byref(out))
        if n == 0 or not out:
            raise RuntimeError("ebcc_encode failed (see the library's log output)")
        try:
            return ctypes.string_at(out.value, n)
        finally:
            self._lib.free_buffer(out)

    def decode(self, buf, out=None):
        raw = bytes(buf)
        src = ctypes.create_string_buffer(raw, len(raw))
        if out is not None:
            target = np.asarray(out)
            if target.dtype != np.float32 or not target.flags.c_contiguous:
                raise TypeError("`out` must be a C-contiguous float32 array")
            ptr = ctypes.c_void_p(target.ctypes.data)                    # the library writes into the caller's buffer
            n = self._lib.ebcc_decode(src, len(raw), ctypes.byref(ptr))
            if n == 0:
                raise RuntimeError("ebcc_decode failed")
            if ptr.value != target.ctypes.data:                          # constant fields come back in a fresh buffer
                target.reshape(-1)[:n] = np.ctypeslib.as_array(ctypes.cast(ptr, ctypes.POINTER(ctypes.c_float)), (n,))
                self._lib.free_buffer(ptr)
            return out
        ptr = ctypes.c_void_p()
        n = self._lib.ebcc_decode(src, len(raw), ctypes.byref(ptr))
        if n == 0 or not ptr:
            raise RuntimeError("ebcc_decode failed")
        try:
            return np.ctypeslib.as_array(ctypes.cast(ptr, ctypes.POINTER(ctypes.c_float)), (n,)).copy()
        finally:
            self._lib.free_buffer(ptr)

    def get_config(self):
        return {"id": self.codec_id, "arglist": [int(v) for v in self.arglist]}

    @classmethod
    def from_config(cls, config):
        return cls(config["arglist"])

    def __repr__(self):
        return f"EBCCZarrFilter(arglist={[int(v) for v in self.arglist]})"


if numcodecs is not None:                                # pragma: no cover
    numcodecs.register_codec(EBCCZarrFilter)
