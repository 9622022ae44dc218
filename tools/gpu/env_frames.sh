#!/bin/bash
# GPU box: environment variants against the batch size.  bash tools/gpu/env_frames.sh "43 85" "A=1" "EBCC_HIP_TRUNC_LEVELS=3" ...
S='import json,sys; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["value"], d["encode_GBps"], d["decode_GBps"])'
FR=$1; shift
for F in $FR; do
  for V in "$@"; do
    echo -n "[frames $F] [$V] "
    env ${V//+/ } timeout -k 10 300 python bench.py --frames $F --steps 4 --warmup 1 --no-cpu-baseline --no-extras 2>/dev/null | tail -1 | python -c "$S"
  done
done
