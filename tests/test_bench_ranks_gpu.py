"""bench.py with more than one rank (the branch the driver's 2 / 4 / 8-GPU scaling runs take: one process per rank, barrier,
max-over-ranks time, per-rank host figures gathered on rank 0, whole-job value) rehearsed on a one-GPU box: two ranks
launched exactly as the driver launches them (python -m torch.distributed.run, 127.0.0.1), sharing the visible device and
talking through gloo (EBCC_BENCH_SHARE_GPU=1: RCCL refuses two ranks on one device).  Every rank codes its own batch with its
own context - frames are independent, there is no collective on the data path (/root/reference/src/ebcc_codec.c:1007-1046 is
a loop over independent chunks)."""
import json
import os
import subprocess
import sys

import pytest

from tests import _lib as L

pytestmark = pytest.mark.gpu


def test_two_ranks_give_one_line_with_the_whole_job_value():
    env = dict(os.environ, EBCC_BENCH_SHARE_GPU="1", MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29517", os.path.join(L.ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--frames", "24"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.stdout[-600:], r.stderr[-1200:])
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, lines                                  # rank 0 alone prints
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 2 and d["scaling"] == "weak"
    assert d["config"]["frames_per_gpu"] == 24
    # whole-job value: both ranks' frames over the slowest rank's time
    assert abs(d["value"] - 2 * 24 * 721 * 1440 * 4 / (d["ms_per_step"] * 1e-3) / 1e9) < 1e-3 * d["value"] + 1e-4
    ranks = d["host"]["ranks"]
    assert len(ranks) == 2 and all(h["pool_threads"] >= 1 for h in ranks)
    # two ranks on one host: each sizes its pool from half the CPUs (LOCAL_WORLD_SIZE, set by the launcher)
    lib = L.product()
    lib.ebcc_hip_host_threads.restype = int
    assert ranks[0]["pool_threads"] <= max(1, lib.ebcc_hip_host_threads(3))
    assert d["max_abs_error"] <= 0.5 * 1.01 + 1e-3
