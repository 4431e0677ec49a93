timeout 900 python -m pytest tests/test_din_gpu.py -x -q -m gpu 2>&1 | tail -2
timeout 300 python tools/din_small_bench.py 2>&1 | tail -3
REC_SMALL_MULTI=0 timeout 300 python tools/din_small_bench.py 2>&1 | tail -3 | sed 's/^/MULTI=0 /'
REC_GEMM_TINY_SPLIT=0 timeout 300 python tools/din_small_bench.py 2>&1 | tail -3 | sed 's/^/TINYSPLIT=0 /'
