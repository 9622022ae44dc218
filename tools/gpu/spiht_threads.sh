#!/bin/bash
# GPU box: k_spiht_encode with 256 / 512 / 1024 list entries per sweep step (rebuilds the library in place on the GPU box's scratch copy and restores the default build at the end)
S='import json,sys; d=json.loads(sys.stdin.read()); k=d["kernels"]; print(d["ms_per_step"], d["value"], d["encode_GBps"], k["spiht_encode"])'
for T in 256 512 1024; do
  touch ebcc_amd/csrc/residual_spiht.hip
  make -C ebcc_amd/csrc EXTRA=-DEBCC_SPIHT_ENC_THREADS=$T -j8 > /dev/null 2>&1 || { echo "build failed for $T"; continue; }
  echo "spiht encode threads $T"
  python -m pytest tests -m gpu -x -q -k "spiht or golden_streams" 2>&1 | tail -1
  for rep in 1 2; do python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras 2>/dev/null | tail -1 | python -c "$S"; done
done
touch ebcc_amd/csrc/residual_spiht.hip
make -C ebcc_amd/csrc -j8 > /dev/null 2>&1 && echo "default build restored"
