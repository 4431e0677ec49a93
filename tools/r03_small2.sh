timeout 900 python -m pytest tests/test_din_gpu.py -x -q -m gpu 2>&1 | tail -3
timeout 300 python tools/din_small_bench.py 2>&1 | tail -3
REC_DIN_HEAD_SMALL=0 timeout 300 python tools/din_small_bench.py 2>&1 | tail -3 | sed 's/^/HEAD_SMALL=0 /'
