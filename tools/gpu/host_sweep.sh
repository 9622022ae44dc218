#!/bin/bash
# GPU box: the host side of the encode - pool width and waiting mode against the container's CPU quota.
#   gpurun --timeout 900 -- 'bash tools/gpu/host_sweep.sh [reps]'
cat /sys/fs/cgroup/cpu.max 2>/dev/null; nproc
S='import json,sys; d=json.loads(sys.stdin.read()); h=d["host"]["ranks"][0]; print(d["ms_per_step"], d["encode_GBps"], d["decode_GBps"], "zstd core-s", h["zstd_core_s_per_step"], "wait ms", h["zstd_wait_ms_per_step"], "proc cpu-s", h["process_cpu_s_per_step"], h.get("cgroup"))'
for rep in $(seq 1 ${1:-2}); do
  for V in "EBCC_HIP_SPIN_SYNC=1 EBCC_HOST_THREADS=64" "EBCC_HOST_THREADS=64" "EBCC_HOST_THREADS=32" "EBCC_HOST_THREADS=24" "EBCC_HOST_THREADS=14" "A=1"; do
    echo -n "[$V] "
    env $V timeout -k 10 300 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-extras 2>/dev/null | tail -1 | python -c "$S"
  done
done
