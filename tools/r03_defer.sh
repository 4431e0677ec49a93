run() { env "$@" timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$*', d['ms_per_step'], round(d['roofline']['in_step_event']['frac'],3), {k: round(v,3) for k,v in d['kernels_ms'].items()})"; }
timeout 600 python -m pytest tests/test_deepfm_gpu.py -x -q -m gpu -n 4 2>&1 | tail -2
for i in 1 2 3; do
run REC_DEEPFM_FOLD_SIDE=1
run REC_DEEPFM_FOLD_SIDE=0
done
