#!/bin/bash
# GPU box: tier-1 / SPIHT decoder kernel time against batch size, state placement and lanes per wave
S='import json,sys; d=json.loads(sys.stdin.read()); k=d["kernels"]; print(d["decode_GBps"], k.get("t1_decode"), k.get("spiht_decode"))'
for F in ${FRAMES:-16 64 256}; do
  for V in ${VARIANTS:-"" "EBCC_HIP_T1_DECODE_GLOBAL=1" "EBCC_T1_LPW=64,64,4,1" "EBCC_T1_LPW=64,64,4,4"}; do
    echo "frames $F [$V]"
    env $V timeout -k 10 300 python bench.py --steps 2 --warmup 1 --frames $F --no-cpu-baseline --no-extras 2>/dev/null | tail -1 | python -c "$S"
  done
done
