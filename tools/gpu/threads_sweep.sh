S='import json,sys; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["value"], d["encode_GBps"], d["decode_GBps"])'
for T in 16 24 32 48; do echo "host threads per slice $T"; EBCC_HOST_THREADS=$T python bench.py --steps 4 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "$S"; done
