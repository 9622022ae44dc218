for l in 64 32 16 8; do
echo "LPW $l: $(EBCC_T1_LPW=$l python bench.py --steps 1 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], {k:v['ms_avg'] for k,v in d['kernels'].items()})" 2>&1 | tail -1)"
done
