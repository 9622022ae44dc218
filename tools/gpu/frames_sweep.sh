#!/bin/bash
# GPU box: per-kernel averages against the batch size (is a kernel bound by its longest chain or by throughput?)
#   gpurun -- 'bash tools/gpu/frames_sweep.sh 32 64 128 256'
S='import json,sys; d=json.loads(sys.stdin.read()); k=d["kernels"]; print(d["ms_per_step"], d["encode_GBps"], d["decode_GBps"], {n: v["ms_avg"] for n, v in k.items()})'
for F in "$@"; do
  echo -n "[frames $F] "
  EBCC_HIP_SLICES=1 timeout -k 10 300 python bench.py --frames $F --steps 4 --warmup 2 --no-cpu-baseline --no-extras 2>/dev/null | tail -1 | python -c "$S"
done
