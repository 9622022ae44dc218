for r in 0 1 2 3 4 5; do
echo "res $r: $(EBCC_DEBUG_RESUME_RES=$r python bench.py --steps 1 --warmup 0 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['kernels']['t1_probe_decode'])" 2>&1 | tail -1)"
done
