"""CPU test of product code: the tier-1 coder (ebcc_amd/csrc/t1_core.hpp, the very source the gfx950 kernels
compile) built for the host and compared block by block with the oracle's tier-1 (random code-blocks, all
sizes/orientations, full and truncated decodes)."""
import os
import subprocess

from tests import _lib as L


def test_t1_core_matches_oracle(tmp_path):
    L.oracle()                                        # make sure libebcc_oracle.so exists
    exe = str(tmp_path / "t1_host_check")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-o", exe, os.path.join(L.ROOT, "tests", "t1_host_check.cpp"),
                           "-L" + os.path.join(L.ROOT, "oracle"), "-lebcc_oracle",
                           "-Wl,-rpath," + os.path.join(L.ROOT, "oracle")])
    out = subprocess.check_output([exe, "250"]).decode()
    assert "0 failures" in out, out
