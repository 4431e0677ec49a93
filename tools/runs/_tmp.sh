cd $GRAFT_REPO_ROOT
ms() { python -c "import sys,json; [print('$1 %.4f ms' % json.loads(l)['ms_per_step']) for l in sys.stdin if l.startswith('{')]"; }
for rep in 1 2; do
for k in 0 4; do
REC_GEMM_DIRECT_KS=$k python bench.py --batch 512 --steps 300 --warmup 30 --no-cpu-baseline --no-other-configs 2>/dev/null | ms "KS=$k 26 tables"
REC_GEMM_DIRECT_KS=$k python bench.py --batch 512 --shared-table --dim 9 --steps 300 --warmup 30 --no-cpu-baseline --no-other-configs 2>/dev/null | ms "KS=$k shared"
done; done
