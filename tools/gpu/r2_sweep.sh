#!/bin/bash
# GPU box: slices and decoder lanes-per-wave with the round-2 kernels
S='import json,sys; d=json.loads(sys.stdin.read()); k=d["kernels"]; print(d["ms_per_step"], d["value"], d["encode_GBps"], d["decode_GBps"], k.get("t1_decode"))'
for SL in 2 3 4 6; do
  echo "slices $SL"
  for rep in 1 2; do EBCC_HIP_SLICES=$SL timeout -k 10 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras 2>/dev/null | tail -1 | python -c "$S"; done
done
for L in 1 2 8; do
  echo "decode lanes per wave $L"
  EBCC_T1_LPW="64,64,4,$L" timeout -k 10 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras 2>/dev/null | tail -1 | python -c "$S"
done
for DS in 1 2 4; do
  echo "decode slices $DS"
  EBCC_HIP_DECODE_SLICES=$DS timeout -k 10 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras 2>/dev/null | tail -1 | python -c "$S"
done
