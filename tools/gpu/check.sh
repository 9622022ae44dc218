#!/bin/bash
# GPU box: parity suite (both tier-1 encoder variants with BOTH=1), bench summary, phase timing.
#   gpurun -- 'bash tools/gpu/check.sh [log-name]'
L=gpurun_out/${1:-check}.log
mkdir -p gpurun_out
{
timeout -k 10 300 python -m pytest tests/test_j2k_gpu.py -m gpu -x -q 2>&1 | tail -5 || exit 1
timeout -k 10 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -5 || exit 1
if [ -n "$BOTH" ]; then EBCC_T1_TWO_PHASE=0 timeout -k 10 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -3 || exit 1; fi
S='import json,sys; d=json.loads(sys.stdin.read()); k=d["kernels"]; print(d["ms_per_step"], d["value"], d["encode_GBps"], d["decode_GBps"], {n:(k[n]["ms_avg"],k[n]["launches"]) for n in k})'
timeout -k 10 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras 2>/dev/null | tail -1 | python -c "$S"
EBCC_HIP_SLICES=1 EBCC_HIP_PHASE_TIMING=1 timeout -k 10 300 python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-extras 2>&1 | grep -E "phase" | tail -16
} 2>&1 | tee $L
