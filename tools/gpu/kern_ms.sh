#!/bin/bash
# GPU box: the bench line's per-kernel averages (live HIP events) for one or more env variants.
#   gpurun -- 'bash tools/gpu/kern_ms.sh [reps] ["VAR=1" ...]'
S='import json,sys; d=json.loads(sys.stdin.read()); k=d["kernels"]; print(d["ms_per_step"], d["encode_GBps"], d["decode_GBps"], {n: v["ms_avg"] for n, v in k.items()})'
REPS=${1:-1}; shift
VARS=("A=1" "$@")
for rep in $(seq 1 $REPS); do
  for V in "${VARS[@]}"; do
    echo -n "[$V] "
    env $V timeout -k 10 300 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-extras 2>/dev/null | tail -1 | python -c "$S"
  done
done
