"""CPU tests of the drop-in boundary: the shared library loads (no GPU needed for dlopen) and exports
every symbol include/*.h declares; struct layouts match the reference's."""
import ctypes
import os
import re
import subprocess

import numpy as np
import pytest

from tests import _lib as L

ROOT = L.ROOT


def _declared(header):
    txt = open(os.path.join(ROOT, "include", header)).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    names = set(re.findall(r"\b(\w+)\s*\(", txt))
    skip = {"defined", "__attribute__", "visibility", "sizeof", "push", "dims", "height"}
    return {n for n in names if n not in skip and not n.isupper()}


@pytest.fixture(scope="module")
def exported():
    if not os.path.exists(L.PRODUCT_SO):
        pytest.skip("library not built (run __graft_entry__.build())")
    out = subprocess.check_output(["nm", "-D", "--defined-only", L.PRODUCT_SO]).decode()
    return {l.split()[-1] for l in out.splitlines() if " T " in l}


def test_reference_api_symbols_exported(exported):
    # /root/reference/src/ebcc_codec.h:41-49, src/h5z_ebcc.c:27-28,38
    for name in ["ebcc_encode", "ebcc_decode", "ebcc_encode_chunking", "ebcc_encode_chunking_compat",
                 "ebcc_decode_chunking", "free_buffer", "print_config", "log_set_level_from_env", "populate_config",
                 "H5PLget_plugin_type", "H5PLget_plugin_info"]:
        assert name in exported, name


def test_every_declared_symbol_is_exported(exported):
    declared = _declared("ebcc_codec.h") | _declared("ebcc_hip.h")
    missing = sorted(n for n in declared if n not in exported)
    assert not missing, missing


def test_config_struct_layout():
    assert ctypes.sizeof(L.CodecConfig) == 64                  # verified sizeof(codec_config_t) on LP64


def test_plugin_descriptor():
    if not os.path.exists(L.PRODUCT_SO):
        pytest.skip("library not built")
    lib = ctypes.CDLL(L.PRODUCT_SO)

    class H5ZClass(ctypes.Structure):                          # /root/reference/src/hdf5_stub.h:34-43
        _fields_ = [("version", ctypes.c_int), ("id", ctypes.c_int), ("encoder_present", ctypes.c_uint),
                    ("decoder_present", ctypes.c_uint), ("name", ctypes.c_char_p), ("can_apply", ctypes.c_void_p),
                    ("set_local", ctypes.c_void_p), ("filter", ctypes.c_void_p)]

    lib.H5PLget_plugin_info.restype = ctypes.POINTER(H5ZClass)
    lib.H5PLget_plugin_type.restype = ctypes.c_int
    assert lib.H5PLget_plugin_type() == 0
    info = lib.H5PLget_plugin_info().contents
    assert (info.version, info.id, info.encoder_present, info.decoder_present) == (1, 308, 1, 1)
    assert info.name == b"HDF5 EBCC filter L&L" and info.filter


def test_populate_config_matches_reference_packing():
    """cd_values layout of /root/reference/ebcc/filter_wrapper.py:22,32-40."""
    if not os.path.exists(L.PRODUCT_SO):
        pytest.skip("library not built")
    import struct
    lib = ctypes.CDLL(L.PRODUCT_SO)
    lib.populate_config.argtypes = [ctypes.POINTER(L.CodecConfig), ctypes.c_size_t, ctypes.POINTER(ctypes.c_uint),
                                    ctypes.c_size_t]
    f2u = lambda v: struct.unpack("<I", struct.pack("<f", v))[0]
    cd = (ctypes.c_uint * 5)(721, 1440, f2u(30.0), 1, f2u(0.5))
    cfg = L.CodecConfig()
    lib.populate_config(ctypes.byref(cfg), 5, cd, 2 * 721 * 1440 * 4)
    assert tuple(cfg.dims) == (2, 721, 1440) and cfg.base_cr == 30.0 and cfg.residual_compression_type == 1
    assert abs(cfg.error - 0.5) < 1e-7 and tuple(cfg.chunk_dims) == (0, 0, 0)


def test_filter_wrapper_mirrors_reference_interface():
    """ebcc_amd.EBCC_Filter against the contract of /root/reference/ebcc/filter_wrapper.py:16-68 (names of the
    residual options, keyword set, attributes, cd_values packing); values cross-checked with the CDO string the
    reference's README quotes for base_cr 30 / max error 0.5 on 721x1440."""
    from ebcc_amd import EBCC_Filter
    f = EBCC_Filter(base_cr=30, height=721, width=1440, residual_opt=("max_error_target", 0.5), data_dim=3)
    assert dict(f) == {"dtype": "float32", "chunks": (1, 721, 1440), "compression": 308,
                       "compression_opts": (721, 1440, 1106247680, 1, 1056964608)}
    assert f.hdf_filter_opts == f["compression_opts"] and f.chunks == (1, 721, 1440) and EBCC_Filter.FILTER_ID == 308
    assert (f.base_cr, f.height, f.width, f.data_dim, f.residual_opt) == (30.0, 721, 1440, 3, ("max_error_target", 0.5))
    assert f.cdo_filter_string() == "308,721,1440,1106247680,1,1056964608"
    assert EBCC_Filter(10, 64, 64, ("relative_error_target", 0.01)).hdf_filter_opts[3:] == (2, 1008981770)
    assert EBCC_Filter(10, 64, 64, ("none", 0)).hdf_filter_opts == (64, 64, 1092616192, 0)
    assert EBCC_Filter(10, 64, 64, None, data_dim=4).chunks == (1, 1, 64, 64)
    assert hash(f) == hash(EBCC_Filter(30, 721, 1440, ("max_error", 0.5)))
    with pytest.raises(ValueError):
        EBCC_Filter(10, 64, 64, ("lossless", 0))


# ---- exit(1) contract of populate_config (/root/reference/src/h5z_ebcc.c:41-92): a fresh child per case
_POPULATE_CHILD = r"""
import ctypes, struct, sys
sys.path.insert(0, {root!r})
from tests import _lib as L
lib = ctypes.CDLL(L.PRODUCT_SO)
lib.populate_config.argtypes = [ctypes.POINTER(L.CodecConfig), ctypes.c_size_t, ctypes.POINTER(ctypes.c_uint), ctypes.c_size_t]
f2u = lambda v: struct.unpack("<I", struct.pack("<f", v))[0]
vals, buf = {vals!r}, {buf!r}
cd = (ctypes.c_uint * max(1, len(vals)))(*[f2u(v) if isinstance(v, float) else v for v in vals])
cfg = L.CodecConfig()
lib.populate_config(ctypes.byref(cfg), len(vals), cd, buf)
print("returned", tuple(cfg.dims))
"""


@pytest.mark.parametrize("name,vals,buf,ok", [
    ("valid", [64, 96, 30.0, 1, 0.5], 64 * 96 * 4, True),
    ("three_values", [64, 96, 30.0], 64 * 96 * 4, False),                     # :41
    ("height_16", [16, 96, 30.0, 0], 16 * 96 * 4, False),                     # :51-57
    ("width_4096", [64, 4096, 30.0, 0], 64 * 4096 * 4, False),
    ("buffer_smaller_than_tile", [64, 96, 30.0, 0], 64 * 95 * 4, False),      # :60-63
    ("buffer_not_divisible", [64, 96, 30.0, 0], 64 * 96 * 4 * 2 + 4, False),  # :64-68
    ("frames_times_height_over_2047", [64, 96, 30.0, 0], 64 * 96 * 4 * 32, False),   # :74-79
    ("mode1_with_4_values", [64, 96, 30.0, 1], 64 * 96 * 4, False),           # :81-92
    ("mode2_with_4_values", [64, 96, 30.0, 2], 64 * 96 * 4, False),
])
def test_populate_config_exit_contract(name, vals, buf, ok):
    import sys
    if not os.path.exists(L.PRODUCT_SO):
        pytest.skip("library not built")
    r = subprocess.run([sys.executable, "-c", _POPULATE_CHILD.format(root=ROOT, vals=vals, buf=buf)], capture_output=True, text=True)
    if ok:
        assert r.returncode == 0 and "returned (1, 64, 96)" in r.stdout, (r.returncode, r.stdout, r.stderr)
    else:
        assert r.returncode == 1 and "returned" not in r.stdout, (name, r.returncode, r.stdout, r.stderr[-300:])


# ---- the host-side codestream parser takes hostile input (the reference leaves that to OpenJPEG)
def test_codestream_parser_rejects_malformed_streams():
    import numpy as np
    if not os.path.exists(L.PRODUCT_SO):
        pytest.skip("library not built")
    lib = ctypes.CDLL(L.PRODUCT_SO)
    lib.ebcc_hip_j2k_parse_check.argtypes = [ctypes.c_char_p, ctypes.c_size_t, ctypes.c_size_t, ctypes.c_size_t]
    lib.ebcc_hip_j2k_parse_check.restype = ctypes.c_int
    h, w = 100, 130
    u16, _, _ = L.scale_u16(L.era5_like(h, w, 5))
    good = L.orc_j2k_encode(u16, 8.0)
    assert lib.ebcc_hip_j2k_parse_check(good, len(good), h, w) == 0
    assert lib.ebcc_hip_j2k_parse_check(good, len(good), h, w + 1) == 1            # another geometry
    r = np.random.default_rng(7)
    verdicts = {0: 0, 1: 0}
    cases = [good[:k] for k in list(range(0, 200)) + [len(good) // 2, len(good) - 3, len(good) - 1]]
    for _ in range(3000):
        b = bytearray(good)
        for _ in range(int(r.integers(1, 4))):
            pos = int(r.integers(0, min(len(b), 600) if r.random() < 0.7 else len(b)))
            b[pos] = int(r.integers(0, 256)) if r.random() < 0.5 else b[pos] ^ (1 << int(r.integers(0, 8)))
        cases.append(bytes(b))
    for k in range(2, 160, 2):                                                    # every marker length field, maximal
        b = bytearray(good); b[k:k + 2] = b"\xff\xff"; cases.append(bytes(b))
    for c in cases:
        v = lib.ebcc_hip_j2k_parse_check(c, len(c), h, w)
        assert v in (0, 1), "parser accepted an out-of-bounds table entry"
        verdicts[v] += 1
    assert verdicts[1] > 200 and verdicts[0] > 0                                  # both verdicts occur; no crash on the way


def test_host_thread_budget_per_rank():
    """The process-wide pool of compressing threads: the CPUs the process may really use (affinity mask cut to the cgroup
    quota), divided by LOCAL_WORLD_SIZE, minus the threads that steer the GPU."""
    import sys
    if not os.path.exists(L.PRODUCT_SO):
        pytest.skip("library not built")
    child = ("import ctypes, os; lib = ctypes.CDLL(%r); lib.ebcc_hip_host_threads.restype = ctypes.c_int; "
             "s = (ctypes.c_double * 8)(); lib.ebcc_hip_host_stats(s, 0); "
             "print(lib.ebcc_hip_host_threads(1), lib.ebcc_hip_host_threads(4), len(os.sched_getaffinity(0)), int(s[0]), s[1])" % L.PRODUCT_SO)
    env = {k: v for k, v in os.environ.items() if k not in ("LOCAL_WORLD_SIZE", "EBCC_HOST_THREADS", "EBCC_HOST_CPU_QUOTA")}

    def run():
        a = subprocess.check_output([sys.executable, "-c", child], env=env).split()
        return int(a[0]), int(a[1]), int(a[2]), int(a[3]), float(a[4])

    def width(cpus_, slices):
        return max(1, min(64, cpus_ - (min(slices, 2) if cpus_ > 4 else 0)))

    env["EBCC_HOST_CPU_QUOTA"] = "0"                                              # no quota: the affinity mask counts
    one, four, cpus, usable, quota = run()
    assert usable == cpus and quota == 0
    assert one == width(cpus, 1) and four == width(cpus, 4)
    env["EBCC_HOST_CPU_QUOTA"] = "16"                                             # the MI355X box: 16-CPU quota on a 256-thread host
    one, four, cpus, usable, quota = run()
    assert usable == min(cpus, 16) and quota == 16.0
    assert four == width(min(cpus, 32), 4)                                        # (bursts up to twice the quota wide)
    env["EBCC_HOST_CPU_QUOTA"] = "1.5"
    one, four, cpus, usable, quota = run()
    assert usable == min(cpus, 2) and one == four == min(cpus, 3)                 # (small shares: nothing set aside)
    env["EBCC_HOST_CPU_QUOTA"] = "0"
    env["LOCAL_WORLD_SIZE"] = "8"
    one, four, cpus, usable, quota = run()
    share = max(1, cpus // 8)
    assert 1 <= one <= share and 1 <= four <= share
    env["EBCC_HOST_THREADS"] = "3"
    assert run()[0] == 3


def test_kernel_arithmetic_identities_hold():
    """ebcc_hip_selfcheck: the fused inverse level maps s in [0, 65535] with fmaf(s, K_hi, s * K_lo) instead of s / 65535.0f."""
    if not os.path.exists(L.PRODUCT_SO):
        pytest.skip("library not built")
    lib = ctypes.CDLL(L.PRODUCT_SO)
    lib.ebcc_hip_selfcheck.restype = ctypes.c_int
    assert lib.ebcc_hip_selfcheck() == 0


def test_prefault_maps_a_destination_array():
    """ebcc_hip_prefault (host only: no device call): a fresh array's pages are touched by several threads, a small or
    null one is left alone."""
    import numpy as np
    if not os.path.exists(L.PRODUCT_SO):
        pytest.skip("library not built")
    lib = ctypes.CDLL(L.PRODUCT_SO)
    lib.ebcc_hip_prefault.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
    lib.ebcc_hip_prefault.restype = ctypes.c_int
    a = np.empty(96 << 20, np.uint8)                                 # above the 64 MB threshold
    assert lib.ebcc_hip_prefault(a.ctypes.data, a.nbytes) == 0
    assert int(a[::4096].max()) == 0                                 # (a zero was written to every page)
    b = np.full(1 << 20, 7, np.uint8)                                # small: nothing is touched
    assert lib.ebcc_hip_prefault(b.ctypes.data, b.nbytes) == 0 and int(b.min()) == 7
    assert lib.ebcc_hip_prefault(None, 0) != 0


def test_decoder_launch_shape_follows_the_batch():
    """ebcc_hip_plan_decode_lanes (host logic of launch_j2k_decode): a small batch decodes one code-block per wave - its
    launch is as long as its longest chain -, a batch that fills the issue slots four per wave with the longest 1/64 in
    pairs; the counts are multiples of their waves' lanes and never exceed the batch."""
    lib = L.product()
    lib.ebcc_hip_plan_decode_lanes.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
    lib.ebcc_hip_plan_decode_lanes.restype = None
    rng = np.random.default_rng(5)

    def plan(frames):
        n = frames * 298
        r = rng.integers(0, 100, n)
        length = np.where(r < 50, 0, np.where(r < 90, rng.integers(1, 1034, n), np.where(r < 99, rng.integers(1034, 2056, n), rng.integers(2056, 2375, n))))
        table = np.zeros((n, 4), np.int32)
        table[:, 1] = length
        table[:, 2] = np.where(length > 0, 13, 0)
        table[:, 3] = np.where(length > 0, 14, 0)
        out = (ctypes.c_int * 4)()
        lib.ebcc_hip_plan_decode_lanes(table.ctypes.data, n, out)
        assert out[0] >= 0 and out[1] >= 0 and out[1] % 2 == 0 and out[2] % 4 == 0 and out[0] + out[1] + out[2] <= n
        return list(out)

    assert plan(43)[3] == 1 and plan(85)[3] == 1
    assert plan(160)[3] == 2
    big = plan(256)
    assert big[3] == 4 and big[0] == 0 and big[1] == 256 * 298 // 64 // 2 * 2
    empty = (ctypes.c_int * 4)()
    lib.ebcc_hip_plan_decode_lanes(np.zeros((64, 4), np.int32).ctypes.data, 64, empty)   # nothing coded: any shape will do
    assert empty[3] in (1, 2, 4)
    lib.ebcc_hip_plan_decode_lanes(None, 0, empty)


def test_slices_follow_the_batch_size(monkeypatch):
    """ebcc_hip_encode_slices_for: a batch below 96 frames runs as one slice, a larger one as three; EBCC_HIP_SLICES decides
    for every batch that is large enough to be cut (four frames per slice at least)."""
    lib = L.product()
    lib.ebcc_hip_encode_slices_for.restype = ctypes.c_int
    lib.ebcc_hip_encode_slices_for.argtypes = [ctypes.c_size_t]
    monkeypatch.delenv("EBCC_HIP_SLICES", raising=False)
    assert [lib.ebcc_hip_encode_slices_for(n) for n in (1, 43, 95, 96, 256)] == [1, 1, 1, 3, 3]
    monkeypatch.setenv("EBCC_HIP_SLICES", "2")
    assert [lib.ebcc_hip_encode_slices_for(n) for n in (1, 7, 8, 43, 256)] == [1, 1, 2, 2, 2]
    monkeypatch.setenv("EBCC_HIP_SLICES", "1")
    assert lib.ebcc_hip_encode_slices_for(256) == 1
