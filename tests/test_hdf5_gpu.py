"""The real HDF5 plugin path (h5py from the image's conda, filter 308 loaded from this build through
HDF5_PLUGIN_PATH) and the direct-chunk batch helper (ebcc_amd/h5_batch.py); the chunk
bytes HDF5 stored are then compared with the oracle's streams."""
import os
import subprocess

import numpy as np
import pytest

from tests import _lib as L

CONDA_PY = "/opt/conda/bin/python3.9"


@pytest.mark.gpu
def test_hdf5_filter_and_direct_chunk_batch(tmp_path):
    if not os.path.exists(CONDA_PY):
        pytest.skip("no interpreter with h5py in this image")
    if subprocess.call([CONDA_PY, "-c", "import h5py"], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL) != 0:
        pytest.skip("h5py not importable")
    env = dict(os.environ, HDF5_PLUGIN_PATH=os.path.join(L.ROOT, "ebcc_amd"), HDF5_USE_FILE_LOCKING="FALSE")
    env.pop("PYTHONPATH", None)
    r = subprocess.run([CONDA_PY, os.path.join(L.ROOT, "tests", "h5_roundtrip.py"), str(tmp_path)], env=env,
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    assert r.stdout.count("OK") == 5, r.stdout
    # what HDF5 stored == the oracle's frame streams for the same frames
    chunks = np.load(tmp_path / "chunks.npy", allow_pickle=True)
    data = np.load(tmp_path / "data.npy")
    cfg = L.make_config((1,) + data.shape[1:], base_cr=20, error=0.1, residual_type=L.MAX_ERROR)
    L.oracle().orc_set_j2k_backend(0)
    for k in (0, 3, 5):
        assert bytes(chunks[k].tobytes()) == L.orc_encode(data[k], cfg), k
    # two-frame chunks: what HDF5 stored == the oracle's stream for the same two frames
    chunks2 = np.load(tmp_path / "chunks2.npy", allow_pickle=True)
    data2 = np.load(tmp_path / "data2.npy")
    cfg2 = L.make_config((2,) + data2.shape[1:], base_cr=10, error=0.05, residual_type=L.MAX_ERROR)
    for i, k in enumerate((0, 2)):
        assert bytes(chunks2[i].tobytes()) == L.orc_encode(data2[k:k + 2], cfg2), k
