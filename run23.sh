python -m pytest tests -m gpu -x -q 2>&1 | tail -3
S='import json,sys; d=json.loads(sys.stdin.read()); k=d["kernels"]; print(d["ms_per_step"], {n:k[n]["ms_avg"] for n in ("t1_encode","t1_probe_decode","t1_decode")})'
for L in "32,32,16,8" "16,32,16,16" "64,32,32,32" "16,32,8,8"; do
echo "LPW $L"; EBCC_T1_LPW=$L python bench.py --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "$S"
done
