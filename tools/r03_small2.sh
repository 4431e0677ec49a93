timeout 600 python -m pytest tests/test_deepfm_gpu.py -x -q -m gpu -k "planned" 2>&1 | tail -12
for f in 1 0 1 0; do REC_STEP_PLAN=$f timeout 300 python bench.py --batch 512 --steps 300 --warmup 30 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('PLAN=$f B512', d['ms_per_step'], d['value'])"; done
REC_GEMM_TINY_SPLIT=0 timeout 300 python bench.py --batch 512 --steps 300 --warmup 30 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('PLAN=1 TINY_SPLIT=0 B512', d['ms_per_step'], d['value'])"
