python bench.py --frames 128 --steps 1 --warmup 0 --no-cpu-baseline 2>&1 | grep "k_rate frame0" | head -30
