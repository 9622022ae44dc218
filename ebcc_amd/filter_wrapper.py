"""h5py keyword helper with the interface of the reference's `ebcc.filter_wrapper.EBCC_Filter`
(/root/reference/ebcc/filter_wrapper.py:16-68): a Mapping to splat into `create_dataset`, carrying

    dtype='float32', chunks=(1, ..., height, width), compression=308,
    compression_opts = cd_values = (height, width, f32bits(base_cr), mode, [f32bits(error)])

with the same constructor arguments, residual option names ("none", "max_error_target",
"relative_error_target") and attributes (`hdf_filter_opts`, `chunks`, `base_cr`, `height`, `width`,
`residual_opt`, `data_dim`, `FILTER_ID`).  Run as a module it prints the `--filter` string for CDO / nccopy like
the reference's command line (`:70-115`).  Differences: the short names "max_error" / "relative_error" are
accepted too, and an unknown residual option raises instead of printing a message and producing a filter
description without a mode.
"""
import struct
import sys
from collections.abc import Mapping

FILTER_ID = 308
_MODES = {"none": 0, "max_error_target": 1, "relative_error_target": 2, "max_error": 1, "relative_error": 2}


def float_to_uint32(v):
    return struct.unpack("<I", struct.pack("<f", float(v)))[0]


class EBCC_Filter(Mapping):
    """`f.create_dataset("t", shape=x.shape, **EBCC_Filter(base_cr=30, height=721, width=1440, data_dim=x.ndim,
    residual_opt=("max_error_target", 0.5)))`"""
    FILTER_ID = FILTER_ID

    def __init__(self, base_cr, height, width, residual_opt=("none", 0), data_dim=2):
        if not (int(height) > 0 and int(width) > 0):
            raise ValueError("height and width must be positive")
        if residual_opt is None:
            residual_opt = ("none", 0)
        mode, value = residual_opt
        if mode not in _MODES:
            raise ValueError(f"unknown residual option {mode!r}: expected 'none', 'max_error_target' or 'relative_error_target'")
        self.base_cr = float(base_cr)
        self.height, self.width = int(height), int(width)
        self.residual_opt = (mode, value)
        self.data_dim = int(data_dim)
        opts = [self.height, self.width, float_to_uint32(self.base_cr), _MODES[mode]]
        if _MODES[mode]:
            if value is None:
                raise ValueError(f"residual option {mode!r} needs an error value")
            opts.append(float_to_uint32(value))
        self.hdf_filter_opts = tuple(opts)
        self.chunks = (1,) * max(self.data_dim - 2, 0) + (self.height, self.width)

    @property
    def _kwargs(self):
        return {"dtype": "float32", "chunks": self.chunks, "compression": self.FILTER_ID, "compression_opts": self.hdf_filter_opts}

    def __hash__(self):
        return hash((self.FILTER_ID, self.hdf_filter_opts))

    def __getitem__(self, k):
        return self._kwargs[k]

    def __iter__(self):
        return iter(self._kwargs)

    def __len__(self):
        return len(self._kwargs)

    def cdo_filter_string(self):
        """`--filter` argument for CDO / nccopy: '308,H,W,...'"""
        return ",".join(str(v) for v in (self.FILTER_ID,) + self.hdf_filter_opts)


def main(argv=None):
    import argparse
    ap = argparse.ArgumentParser(description="print the HDF5 filter description of an EBCC configuration")
    ap.add_argument("-b", "--base_cr", type=float, default=200, help="base compression ratio")
    ap.add_argument("-H", "--height", type=int, default=721, help="height of the data slice (latitude)")
    ap.add_argument("-W", "--width", type=int, default=1440, help="width of the data slice (longitude)")
    ap.add_argument("-m", "--max_error_target", type=float, default=None, help="max error target")
    ap.add_argument("-r", "--relative_error_target", type=float, default=None, help="relative error target")
    ap.add_argument("--help-cdo", action="store_true", help="print the CDO command line")
    a = ap.parse_args(argv)
    if a.max_error_target:
        opt = ("max_error_target", a.max_error_target)
    elif a.relative_error_target:
        opt = ("relative_error_target", a.relative_error_target)
    else:
        print("Using default settings: relative error target of 0.01", file=sys.stderr)
        opt = ("relative_error_target", 0.01)
    flt = EBCC_Filter(base_cr=a.base_cr, height=a.height, width=a.width, residual_opt=opt)
    print(f"Base compression ratio: {a.base_cr}; HeightxWidth: {a.height}x{a.width}; residual option: {opt[0]}, {opt[1]}", file=sys.stderr)
    if a.help_cdo:
        print(f"cdo -b F32 -f nc4 --filter {flt.cdo_filter_string()} copy original.nc compressed.nc", file=sys.stderr)
        print(f"(the chunk size of original.nc has to be a multiple of the tile size {a.height}x{a.width})", file=sys.stderr)
    print(flt.cdo_filter_string())


if __name__ == "__main__":
    main()
