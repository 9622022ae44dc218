EBCC_HIP_PHASE_TIMING=1 python bench.py --steps 1 --warmup 1 --no-cpu-baseline 2>&1 | grep -E "phase" | tail -12
