timeout 1500 python -m pytest tests -x -q -m gpu -n 4 2>&1 | tail -3
for f in 1 0 1 0; do echo "DROP=$f $(REC_GROUP_DROP=$f timeout 300 python tools/slot_dnn_bench.py --opt ps 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['train_step_ms'], d['kernels_ms'])")"; done
for f in 1 0; do REC_GROUP_DROP=$f timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('DROP=$f deepfm', d['ms_per_step'])"; done
