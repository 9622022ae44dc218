#!/bin/bash
# GPU box: tier-1 decoder, share of code-blocks at few lanes per wave against lanes per wave of the rest
S='import json,sys; d=json.loads(sys.stdin.read()); k=d["kernels"]; print(d["decode_GBps"], k.get("t1_decode"), k.get("spiht_decode"))'
for V in ${VARIANTS:-"EBCC_T1_DEC_MIX=1,2" "EBCC_T1_DEC_MIX=16,4" "EBCC_T1_DEC_MIX=32,4" "EBCC_T1_DEC_MIX=64,4"}; do
  echo "[$V]"
  for rep in 1 2; do
    env ${V//+/ } timeout -k 10 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras 2>/dev/null | tail -1 | python -c "$S"
  done
done
