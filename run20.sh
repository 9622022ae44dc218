python -m pytest tests -m gpu -x -q 2>&1 | tail -3
S='import json,sys; d=json.loads(sys.stdin.read()); print({k:d[k] for k in ("value","ms_per_step","encode_GBps","decode_GBps","kernels")})'
python bench.py --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "$S"
EBCC_HIP_PURE_SEARCH=concurrent python bench.py --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "$S"
EBCC_HIP_PHASE_TIMING=1 python bench.py --steps 1 --warmup 1 --no-cpu-baseline 2>&1 | grep -E "phase" | tail -9
