S='import json,sys; d=json.loads(sys.stdin.read()); k=d["kernels"]; print(d["ms_per_step"], d["value"], d["encode_GBps"], d["decode_GBps"], {n:(k[n]["ms_avg"],k[n]["launches"]) for n in k})'
for K in 4 2 3; do echo "slices $K"; EBCC_HIP_SLICES=$K python bench.py --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "$S"; done
echo "slices 4 + GPU_MAX_HW_QUEUES=8"; GPU_MAX_HW_QUEUES=8 EBCC_HIP_SLICES=4 python bench.py --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "$S"
EBCC_HIP_SLICES=4 EBCC_HIP_PHASE_TIMING=1 python bench.py --steps 1 --warmup 1 --no-cpu-baseline 2>&1 | grep -E "phase" | tail -36
