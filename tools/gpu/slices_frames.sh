#!/bin/bash
# GPU box: encode slices against the batch size.  bash tools/gpu/slices_frames.sh "16 43 85 128" "1 2 3"
S='import json,sys; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["value"], d["encode_GBps"], d["decode_GBps"])'
for F in $1; do
  for K in $2; do
    echo -n "[frames $F] [slices $K] "
    EBCC_HIP_SLICES=$K timeout -k 10 300 python bench.py --frames $F --steps 4 --warmup 1 --no-cpu-baseline --no-extras 2>/dev/null | tail -1 | python -c "$S"
  done
done
