# cost of the host entropy stage by zstd level (streams stay decodable, only level 22 is byte-identical to the reference)
S='import json,sys; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["value"], d["encode_GBps"], d["compressed_bytes_per_frame"])'
for L in 22 19 15 9 3 1; do echo "zstd level $L"; EBCC_ZSTD_LEVEL=$L python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras 2>/dev/null | tail -1 | python -c "$S"; done
