timeout 600 python -m pytest tests/test_gemm_gpu.py -x -q -m gpu -k "head" 2>&1 | tail -2
for f in 1 0 1 0; do REC_MLP_HEAD_FUSED=$f timeout 300 python bench.py --steps 30 --warmup 5 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('HEAD=$f', d['value'], d['ms_per_step'], d['roofline']['frac'])"; done
