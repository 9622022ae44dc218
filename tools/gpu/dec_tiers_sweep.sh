#!/bin/bash
# GPU box: the tier-1 decoder's lanes per wave against the batch size.  bash tools/gpu/dec_tiers_sweep.sh "43 85 128 256" "A=1" "EBCC_T1_LPW=64,64,4,1" ...
S='import json,sys; d=json.loads(sys.stdin.read()); k=d["kernels"]; print(d["ms_per_step"], d["decode_GBps"], "t1_decode", k["t1_decode"]["ms_avg"])'
FR=$1; shift
for F in $FR; do
  for V in "$@"; do
    echo -n "[frames $F] [$V] "
    env ${V//+/ } timeout -k 10 300 python bench.py --frames $F --steps 3 --warmup 1 --no-cpu-baseline --no-extras 2>/dev/null | tail -1 | python -c "$S"
  done
done
