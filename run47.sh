S='import json,sys; d=json.loads(sys.stdin.read()); k=d["kernels"]; print(d["ms_per_step"], d["value"], d["encode_GBps"], d["decode_GBps"])'
echo "decode slices 2 (default)"; python bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "$S"
echo "decode slices 1 + concurrent residual"; EBCC_HIP_DECODE_SLICES=1 EBCC_HIP_CONCURRENT_RESIDUAL_DECODE=1 python bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "$S"
echo "decode slices 1 serial"; EBCC_HIP_DECODE_SLICES=1 python bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "$S"
echo "decode slices 2 + concurrent residual"; EBCC_HIP_CONCURRENT_RESIDUAL_DECODE=1 python bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "$S"
