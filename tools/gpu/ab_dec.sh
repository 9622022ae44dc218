#!/bin/bash
# GPU box: decode rate of several settings, alternating (same box)
S='import json,sys; d=json.loads(sys.stdin.read()); print(d["decode_GBps"], d["encode_GBps"], d["ms_per_step"])'
for rep in 1 2 3 4; do
  for V in "EBCC_HIP_DECODE_SLICES=2 EBCC_T1_DEC_MIX=1,2" "EBCC_HIP_DECODE_SLICES=1 EBCC_T1_DEC_MIX=32,4" "EBCC_HIP_DECODE_SLICES=1 EBCC_T1_DEC_MIX=1,2" "EBCC_HIP_DECODE_SLICES=2 EBCC_T1_DEC_MIX=4,4"; do
    echo -n "[$V] "
    env $V timeout -k 10 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras 2>/dev/null | tail -1 | python -c "$S"
  done
done
