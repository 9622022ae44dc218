S='import json,sys; d=json.loads(sys.stdin.read()); k=d["kernels"]; print(d["ms_per_step"], d["value"], d["encode_GBps"], d["decode_GBps"], {n:(k[n]["ms_avg"],k[n]["launches"]) for n in k})'
for K in 2 3 4 1; do echo "slices $K"; EBCC_HIP_SLICES=$K python bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "$S"; done
