"""ebcc_amd - MI355X (gfx950) build of the EBCC error-bounded compressor.

Mirrors the reference package's discovery contract (/root/reference/ebcc/__init__.py:5-29): the HDF5 filter
plugin `libh5z_ebcc.so` lives next to this file and `EBCC_FILTER_PATH` / `EBCC_FILTER_DIR` point at it, so
`HDF5_PLUGIN_PATH=ebcc_amd.EBCC_FILTER_DIR` makes h5py/netCDF/CDO load filter 308 from here.
There is no CPU fallback: if the library has not been built, importing the names below raises.
"""
import glob
import os

_HERE = os.path.dirname(os.path.abspath(__file__))


def _find():
    hits = sorted(glob.glob(os.path.join(_HERE, "libh5z_ebcc*.so")))
    if not hits:
        raise FileNotFoundError(
            f"libh5z_ebcc.so not found in {_HERE}: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "or `make -C ebcc_amd/csrc` (hipcc, --offload-arch=gfx950)")
    return hits[0]


def __getattr__(name):
    if name == "EBCC_FILTER_PATH":
        return _find()
    if name == "EBCC_FILTER_DIR":
        return os.path.dirname(_find())
    raise AttributeError(name)


def load():
    """ctypes handle of the plugin / C-ABI library."""
    import ctypes
    return ctypes.CDLL(_find())


from .filter_wrapper import EBCC_Filter  # noqa: E402,F401
