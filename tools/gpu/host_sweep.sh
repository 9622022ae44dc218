#!/bin/bash
# GPU box: the host side of the encode - pool width, waiting mode, entropy-stage shortcut and early-exit probes against the
# container's CPU quota (alternating runs of one binary).   gpurun --timeout 900 -- 'bash tools/gpu/host_sweep.sh [reps] ["VAR=1" ...]'
cat /sys/fs/cgroup/cpu.max 2>/dev/null; nproc
S='import json,sys; d=json.loads(sys.stdin.read()); h=d["host"]["ranks"][0]; print(d["ms_per_step"], d["encode_GBps"], d["decode_GBps"], "zstd core-s", h["zstd_core_s_per_step"], "wait ms", h["zstd_wait_ms_per_step"], "MB", h["zstd_MB_per_step"], "skipped MB", h.get("prefix_MB_per_step_decided_without_zstd"), "proc cpu-s", h["process_cpu_s_per_step"], h.get("cgroup"))'
REPS=${1:-2}; shift
VARS=("A=1" "$@")
for rep in $(seq 1 $REPS); do
  for V in "${VARS[@]}"; do
    echo -n "[$V] "
    env $V timeout -k 10 300 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-extras 2>/dev/null | tail -1 | python -c "$S"
  done
done
