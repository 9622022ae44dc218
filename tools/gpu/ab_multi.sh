#!/bin/bash
# GPU box: several settings on the default bench, alternating (same box).  bash tools/gpu/ab_multi.sh REPS "ENV1" "ENV2" ...   ('+' joins variables)
S='import json,sys; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["value"], d["encode_GBps"], d["decode_GBps"])'
R=$1; shift
for rep in $(seq 1 $R); do
  for V in "$@"; do
    echo -n "[$V] "
    env ${V//+/ } timeout -k 10 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras 2>/dev/null | tail -1 | python -c "$S"
  done
done
