"""Multi-GPU sharding of a frame stack: frames (EBCK chunks / HDF5 chunks) are independent, so each rank of a
`torch.distributed` job codes a contiguous block of frames on its own GPU and rank 0 concatenates the compressed
streams on the host in chunk order.  No collective touches the data path; the only exchange is the gather of the
(tiny) compressed streams, done here with `gather_object` on whatever backend the group uses (RCCL/gloo).

The container is the reference's EBCK layout (/root/reference/src/ebcc_codec.c:204-213,1007-1046):
80-byte header, then per chunk `u64 nbytes | EBCC stream`, chunks in C order of chunk index.
"""
import struct

EBCK_MAGIC = b"EBCK"


def shard_range(n_items, rank, world):
    """Contiguous block of ceil(n/world) items per rank (keeps output order == input order)."""
    per = -(-n_items // world)
    lo = min(rank * per, n_items)
    return lo, min(lo + per, n_items)


def ebck_header(dims, chunk_dims):
    counts = [-(-d // c) for d, c in zip(dims, chunk_dims)]
    num_chunks = counts[0] * counts[1] * counts[2]
    chunk_size = chunk_dims[0] * chunk_dims[1] * chunk_dims[2]
    return struct.pack("<4sIII3Q3QQQ", EBCK_MAGIC, 1, 3, 0, *dims, *chunk_dims, num_chunks, chunk_size)


def assemble_ebck(dims, chunk_dims, streams):
    """streams: EBCC frame streams in C order of chunk index."""
    out = [ebck_header(dims, chunk_dims)]
    for s in streams:
        out.append(struct.pack("<Q", len(s)))
        out.append(bytes(s))
    return b"".join(out)


def gather_streams(local_streams, group=None, dst=0):
    """Ordered host-side concatenation of per-rank stream lists on rank `dst` (None elsewhere)."""
    import torch.distributed as dist
    if not dist.is_available() or not dist.is_initialized() or dist.get_world_size(group) == 1:
        return list(local_streams)
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    gathered = [None] * world if rank == dst else None
    dist.gather_object(list(local_streams), gathered, dst=dst, group=group)
    if rank != dst:
        return None
    return [s for part in gathered for s in part]


def encode_stack_sharded(frames, config, encode_fn, group=None):
    """frames: (n, H, W) array holding the WHOLE stack on every rank (or at least this rank's block);
    encode_fn(block, config) -> list of EBCC streams (e.g. tests/_lib.Context.encode_frames on the rank's GPU).
    Returns the EBCK container on rank 0, None elsewhere."""
    import torch.distributed as dist
    n, h, w = frames.shape
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    lo, hi = shard_range(n, rank, world)
    local = encode_fn(frames[lo:hi], config) if hi > lo else []
    streams = gather_streams(local, group)
    if streams is None:
        return None
    return assemble_ebck((n, h, w), (1, h, w), streams)


def host_threads_per_rank(lib, slices=2):
    """Compressing (zstd) host threads of this rank: ONE pool per process, shared by all slices of all calls.  The library
    sizes it from the CPUs the process may really use (affinity mask cut down to the cgroup CPU quota) divided by
    LOCAL_WORLD_SIZE (set by torchrun), so that the ranks of a node share its cores - ebcc_hip_host_threads()."""
    lib.ebcc_hip_host_threads.restype = __import__("ctypes").c_int
    return int(lib.ebcc_hip_host_threads(int(slices)))
