S='import json,sys; d=json.loads(sys.stdin.read()); k=d["kernels"]; print(d["ms_per_step"], d["value"], d["encode_GBps"], d["decode_GBps"], {n:(k[n]["ms_avg"],k[n]["launches"]) for n in k if n.startswith("t1_")})'
for L in ${LPWS:-64,64,16,8 32,64,16,8 16,64,16,8}; do echo "LPW $L"; EBCC_T1_LPW=$L python bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "$S"; done
