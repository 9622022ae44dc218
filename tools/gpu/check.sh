#!/bin/bash
# GPU box: parity suite (both tier-1 encoder variants), bench summary, phase timing.   gpurun -- 'bash tools/gpu/check.sh'
python -m pytest tests -m gpu -x -q 2>&1 | tail -3
[ -n "$BOTH" ] && EBCC_T1_TWO_PHASE=0 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
S='import json,sys; d=json.loads(sys.stdin.read()); k=d["kernels"]; print(d["ms_per_step"], d["value"], d["encode_GBps"], d["decode_GBps"], {n:(k[n]["ms_avg"],k[n]["launches"]) for n in k})'
python bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "$S"
EBCC_HIP_SLICES=1 EBCC_HIP_PHASE_TIMING=1 python bench.py --steps 1 --warmup 1 --no-cpu-baseline 2>&1 | grep -E "phase" | tail -9
