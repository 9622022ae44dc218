python -m pytest tests/test_j2k_gpu.py -m gpu -x -q 2>&1 | grep -E "rate check|passed|failed" | sort | uniq -c | head -20
