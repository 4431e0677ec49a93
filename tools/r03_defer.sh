run() { env "$@" timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$*', d['ms_per_step'], {k: round(v,3) for k,v in d['kernels_ms'].items()})"; }
timeout 900 python -m pytest tests/test_gemm_gpu.py tests/test_deepfm_gpu.py -x -q -m gpu -n 4 2>&1 | tail -2
for i in 1 2 3; do
run REC_GEMM_144=1
run REC_GEMM_144=0
done
run REC_GEMM_144=1 REC_DW0_SPLIT=24
run REC_GEMM_144=1 REC_DW0_SPLIT=0
