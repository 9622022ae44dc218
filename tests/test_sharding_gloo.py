"""N>1 path on CPU: two gloo ranks shard a frame stack, code their blocks, rank 0 concatenates.  The result
must be byte-identical to the serial EBCK container (oracle restatement of ebcc_encode_chunking).  The per-frame
encoder here is the CPU oracle standing in for the GPU (no GPU in this tier); the partition, ordering and
container assembly are the product code under test (ebcc_amd/sharding.py)."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_frames, q):
    import torch.distributed as dist
    from ebcc_amd import sharding
    from tests import _lib as L
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    frames = np.stack([L.era5_like(32, 40, 100 + i) for i in range(n_frames)])
    cfg = L.make_config((1, 32, 40), base_cr=10.0, error=0.05, residual_type=L.MAX_ERROR)
    L.oracle().orc_set_j2k_backend(0)

    def encode_fn(block, config):
        return [L.orc_encode(f, config) for f in block]

    out = sharding.encode_stack_sharded(frames, cfg, encode_fn)
    if rank == 0:
        # the host-thread budget of a rank (product library; no GPU needed): with LOCAL_WORLD_SIZE ranks on the node a rank
        # keeps its compressing threads - all slices together - within its share of the CPUs
        import ctypes
        budget = None
        if os.path.exists(L.PRODUCT_SO):
            os.environ["LOCAL_WORLD_SIZE"] = "8"
            os.environ.pop("EBCC_HOST_THREADS", None)
            budget = (sharding.host_threads_per_rank(ctypes.CDLL(L.PRODUCT_SO), 2), len(os.sched_getaffinity(0)))
        q.put((out, budget))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n_frames", [5, 2, 1])
def test_two_rank_sharded_container_equals_serial(n_frames):
    import torch.multiprocessing as mp
    from tests import _lib as L
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n_frames, q)) for r in range(2)]
    for p in procs:
        p.start()
    got, budget = q.get(timeout=120)
    if budget is not None:
        threads, cpus = budget
        assert 1 <= threads <= max(1, cpus // 8), budget          # (one pool per process, within the rank's share)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    frames = np.stack([L.era5_like(32, 40, 100 + i) for i in range(n_frames)])
    cfg = L.make_config((n_frames, 32, 40), (1, 32, 40), base_cr=10.0, error=0.05, residual_type=L.MAX_ERROR)
    L.oracle().orc_set_j2k_backend(0)
    want = L.orc_encode(frames, cfg, "orc_ebcc_encode_chunking")
    assert got == want
    dec = L.orc_decode(got, "orc_ebcc_decode_chunking").reshape(frames.shape)
    assert np.abs(dec - frames).max() <= 0.05 * 1.01


def test_shard_ranges_cover_everything_in_order():
    from ebcc_amd.sharding import shard_range
    for n in (0, 1, 7, 8, 9, 256, 32768):
        for world in (1, 2, 4, 8):
            spans = [shard_range(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
