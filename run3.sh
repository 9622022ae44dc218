python -m pytest tests/test_codec_gpu.py -m gpu -x -q 2>&1 | tail -40
