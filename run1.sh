set -x
ls /opt/conda/lib/libopenjp2.so.7 /opt/conda/lib/libzstd.so.1 /opt/conda/bin/python3.9 2>&1
rocminfo | grep -E "gfx|Compute Unit" | head -4
nproc; free -g | head -2
python -m pytest tests/test_residual_gpu.py -m gpu -x -q 2>&1 | tail -30
