#!/bin/bash
# GPU box: A/B of an environment switch on the default bench (same box, alternating runs)
#   gpurun -- 'bash tools/gpu/ab.sh "EBCC_HIP_SPECULATION=1" [reps]'
S='import json,sys; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["value"], d["encode_GBps"], d["decode_GBps"])'
for rep in $(seq 1 ${2:-3}); do
  for V in "A=1" "$1"; do
    echo -n "[$V] "
    env $V timeout -k 10 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras 2>/dev/null | tail -1 | python -c "$S"
  done
done
