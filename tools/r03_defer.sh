run() { env "$@" timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$*', d['ms_per_step'], round(d['mlp_gemm']['frac'],3), round(d['roofline']['in_step_event']['frac'],3))"; }
for i in 1 2 3; do
run REC_MLP_DW_STREAM=0 REC_DEEPFM_GROUP_AT=bwd
run REC_MLP_DW_STREAM=0 REC_DEEPFM_GROUP_AT=fwd
done
