run() { env "$@" timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --force-sharded $EXTRA 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$* $EXTRA', round(d['ms_per_step'],4))"; }
for c in 192 0 192 0 160 224; do run REC_SHARD_DW_CUS=$c; done
EXTRA="--table ps --hashed-rows 1250000000"
for c in 192 0 192 0; do run REC_SHARD_DW_CUS=$c; done
