S='import json,sys; d=json.loads(sys.stdin.read()); k=d["kernels"]; print(d["ms_per_step"], d["value"], d["encode_GBps"], d["decode_GBps"], {n:(k[n]["ms_avg"],k[n]["launches"]) for n in k})'
for i in 1 2 3; do echo "slices 2 run $i"; EBCC_HIP_SLICES=2 python bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "$S"; done
