run() { env "$@" timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$*', d['ms_per_step'], round(d['mlp_gemm']['frac'],3), {k: round(v,3) for k,v in d['kernels_ms'].items()})"; }
for i in 1 2; do
run REC_GEMM_SKIP_REDUCE=0
run REC_GEMM_SKIP_REDUCE=1
done
