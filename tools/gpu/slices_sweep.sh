S='import json,sys; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["value"], d["encode_GBps"], d["decode_GBps"])'
for K in ${KS:-2 3 4}; do for Q in ${QS:-4 8}; do echo "slices $K hw queues $Q zstd ${EBCC_ZSTD_LEVEL:-22}"; GPU_MAX_HW_QUEUES=$Q EBCC_HIP_SLICES=$K python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras 2>/dev/null | tail -1 | python -c "$S"; done; done
