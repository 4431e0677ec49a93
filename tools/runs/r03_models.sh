mkdir -p gpurun_out/r03_models
timeout 1200 python tools/bench_models.py > gpurun_out/r03_models/models.txt 2>&1
tail -40 gpurun_out/r03_models/models.txt | cut -c1-260
