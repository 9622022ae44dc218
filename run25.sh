S='import json,sys; d=json.loads(sys.stdin.read()); k=d["kernels"]; print(d["ms_per_step"], d["value"], {n:k[n]["ms_avg"] for n in ("t1_encode","t1_probe_decode","t1_decode","rate_alloc","spiht_encode")})'
for F in 32 64 128 256; do echo "frames $F"; python bench.py --frames $F --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "$S"; done
