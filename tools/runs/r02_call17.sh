#!/bin/bash
# round 2, GPU call 17: gpubox GPU test; where does the host block in the world-1 sharded step (cProfile of bench.py)
mkdir -p gpurun_out/r02_call17
o=gpurun_out/r02_call17
timeout 600 python -m pytest tests/test_gpubox.py tests/test_xdeepfm.py -x -q -m gpu > $o/pytest.txt 2>&1; echo "pytest rc=$?" >> $o/pytest.txt
tail -5 $o/pytest.txt
timeout 600 python -m cProfile -o $o/sharded.prof bench.py --force-sharded --no-cpu-baseline --steps 60 --warmup 10 > $o/bench_sh.json 2> $o/bench_sh.err
python - <<'PY' > gpurun_out/r02_call17/sharded_profile.txt
import pstats
p = pstats.Stats("gpurun_out/r02_call17/sharded.prof")
p.sort_stats("tottime").print_stats(28)
p.sort_stats("cumulative").print_stats(45)
PY
head -60 $o/sharded_profile.txt | cut -c1-200
rm -f $o/sharded.prof
