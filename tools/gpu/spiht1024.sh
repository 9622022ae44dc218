#!/bin/bash
# GPU box: the SPIHT encoder with 1024 list entries per sweep (parity + bench), then the default build again
S='import json,sys; d=json.loads(sys.stdin.read()); k=d["kernels"]; print(d["ms_per_step"], d["value"], d["encode_GBps"], k["spiht_encode"])'
for T in 1024 512; do
  touch ebcc_amd/csrc/residual_spiht.hip
  make -C ebcc_amd/csrc EXTRA=-DEBCC_SPIHT_ENC_THREADS=$T -j8 > /dev/null 2>&1 || { echo "build failed for $T"; continue; }
  echo "spiht encode threads $T"
  timeout -k 10 600 python -m pytest tests/test_residual_gpu.py tests/test_codec_gpu.py -m gpu -x -q -k "streams_bit_exact or dense or golden_streams" 2>&1 | tail -2
  for rep in 1 2; do timeout -k 10 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras 2>/dev/null | tail -1 | python -c "$S"; done
done
touch ebcc_amd/csrc/residual_spiht.hip
make -C ebcc_amd/csrc -j8 > /dev/null 2>&1 && echo "default build restored"
