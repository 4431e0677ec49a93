timeout 900 python -m pytest tests/test_deepfm_gpu.py -x -q -m gpu -k planned 2>&1 | tail -2
