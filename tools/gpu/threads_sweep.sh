# GPU box: host threads per encode call for the zstd stage (EBCC_HOST_THREADS), two rounds to see the run-to-run spread
S='import json,sys; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["value"], d["encode_GBps"], d["decode_GBps"])'
for rep in 1 2; do for T in 8 16 24 32 48 64; do echo "host threads per slice $T"; EBCC_HOST_THREADS=$T python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras 2>/dev/null | tail -1 | python -c "$S"; done; done
