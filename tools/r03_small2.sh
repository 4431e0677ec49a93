timeout 900 python -m pytest tests/test_deepfm_gpu.py -x -q -m gpu -k "small_merge or planned" 2>&1 | tail -3
