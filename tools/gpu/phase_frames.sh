#!/bin/bash
# GPU box: phase times of ONE slice alone against its size (what a slice costs without the others beside it).
#   gpurun -- 'bash tools/gpu/phase_frames.sh 43 85 128 256'
for F in "$@"; do
  EBCC_HIP_SLICES=1 EBCC_HIP_PHASE_TIMING=1 timeout -k 10 300 python bench.py --frames $F --steps 3 --warmup 1 --no-cpu-baseline --no-extras 2> gpurun_out/phase_frames.err | tail -1 | python -c "
import json,sys,re,collections
d=json.loads(sys.stdin.read())
acc=collections.defaultdict(list)
order=[]
for l in open('gpurun_out/phase_frames.err'):
    m=re.match(r'ebcc-mi355x phase (.+?)\s+([0-9.]+) ms', l)
    if m:
        k=m.group(1).strip()
        if k not in acc: order.append(k)
        acc[k].append(float(m.group(2)))
print('[frames $F]', d['ms_per_step'], d['encode_GBps'], d['decode_GBps'])
for k in order: print('   %-50s %.1f' % (k, sum(acc[k][-3:])/max(1,len(acc[k][-3:]))))
"
done
