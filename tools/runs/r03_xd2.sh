timeout 900 python -m pytest tests/test_xdeepfm.py -x -q -m gpu 2>&1 | tail -3
for w in 1 0 1; do echo "REC_CIN_PAD=$w $(REC_CIN_PAD=$w timeout 300 python tools/xdeepfm_chunk_bench.py 4096 2>&1 | tail -1)"; done
