timeout 1500 python -m pytest tests -x -q -m gpu -n 4 2>&1 | tail -3
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --force-sharded 2>&1 | tail -1 | cut -c1-700
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --force-sharded --table ps --hashed-rows 1250000000 2>&1 | tail -1 | cut -c1-700
