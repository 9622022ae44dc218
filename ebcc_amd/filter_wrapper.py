"""h5py keyword helper with the same packing as /root/reference/ebcc/filter_wrapper.py:16-68:
cd_values = (frame_rows, frame_cols, f32bits(base_cr), mode, [f32bits(error)])."""
import struct
from collections.abc import Mapping

FILTER_ID = 308
_MODES = {"none": 0, "max_error": 1, "relative_error": 2}


def float_to_uint32(v):
    return struct.unpack("<I", struct.pack("<f", float(v)))[0]


class EBCC_Filter(Mapping):
    """`f.create_dataset(..., **EBCC_Filter(base_cr=30, height=721, width=1440, residual_opt=("max_error", 0.5)))`"""

    def __init__(self, base_cr, height, width, data_dim=None, residual_opt=("none", None), filter_id=FILTER_ID):
        mode, value = residual_opt if residual_opt is not None else ("none", None)
        if mode not in _MODES:
            raise ValueError(f"unknown residual mode {mode!r}; expected one of {sorted(_MODES)}")
        opts = [int(height), int(width), float_to_uint32(base_cr), _MODES[mode]]
        if _MODES[mode]:
            if value is None:
                raise ValueError(f"residual mode {mode!r} needs an error value")
            opts.append(float_to_uint32(value))
        self._kw = {"compression": filter_id, "compression_opts": tuple(opts)}

    def __getitem__(self, k):
        return self._kw[k]

    def __iter__(self):
        return iter(self._kw)

    def __len__(self):
        return len(self._kw)

    def cdo_filter_string(self):
        """`--filter` argument for CDO / nccopy: '308,H,W,...'"""
        return ",".join(str(v) for v in (self._kw["compression"],) + self._kw["compression_opts"])
