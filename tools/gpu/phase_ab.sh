#!/bin/bash
# GPU box: mean host-side phase times of a slice (EBCC_HIP_PHASE_TIMING) under environment variants, alternating runs.
#   gpurun -- 'bash tools/gpu/phase_ab.sh REPS "A=1" "VAR=1" ...'   ('+' joins variables)
R=$1; shift
for rep in $(seq 1 $R); do
  for V in "$@"; do
    env ${V//+/ } EBCC_HIP_PHASE_TIMING=1 timeout -k 10 300 python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-extras 2> gpurun_out/phase_ab.err | tail -1 | python -c "
import json,sys,re,collections
d=json.loads(sys.stdin.read())
acc=collections.defaultdict(list)
for l in open('gpurun_out/phase_ab.err'):
    m=re.match(r'ebcc-mi355x phase (.+?)\s+([0-9.]+) ms', l)
    if m: acc[m.group(1).strip()].append(float(m.group(2)))
keys=['analysis (dwt, tier-1, ckpt)','first probe','rate search 1','tails + residual range','residual: analysis, SPIHT, whole-stream probe','truncation search','zstd: wait for the workers','decode: kernels']
print('[$V]', d['ms_per_step'], ' '.join('%s=%.1f' % (k.split(' ')[0][:9], sum(acc[k][-12:])/max(1,len(acc[k][-12:]))) for k in keys))
"
  done
done
