timeout 900 python -m pytest tests/test_deepfm_gpu.py -x -q -m gpu -n 4 2>&1 | tail -2
for f in 16 0 16 0; do REC_DW0_SPLIT=$f timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('DW0_SPLIT=$f', d['ms_per_step'], d['value'])"; done
