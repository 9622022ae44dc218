#!/bin/bash
# GPU box: k_rate with 320 / 384 / 512 threads per frame (rebuilds the library in place on the GPU box's scratch copy and restores the default build at the end)
S='import json,sys; d=json.loads(sys.stdin.read()); k=d["kernels"]; print(d["ms_per_step"], d["value"], d["encode_GBps"], k["rate_alloc"])'
for T in 512 384 320 256; do
  touch ebcc_amd/csrc/j2k_rate.hip
  make -C ebcc_amd/csrc EXTRA=-DEBCC_RATE_THREADS=$T -j8 > /dev/null 2>&1 || { echo "build failed for $T"; continue; }
  echo "k_rate threads $T"
  python -m pytest tests -m gpu -x -q -k "golden_streams" 2>&1 | tail -1
  for rep in 1 2; do python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras 2>/dev/null | tail -1 | python -c "$S"; done
done
touch ebcc_amd/csrc/j2k_rate.hip
make -C ebcc_amd/csrc -j8 > /dev/null 2>&1 && echo "default build restored"
