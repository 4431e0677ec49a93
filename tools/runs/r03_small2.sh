timeout 900 python -m pytest tests/test_dcn_v2_gpu.py -x -q -m gpu 2>&1 | tail -3
for f in 1 0; do REC_STEP_PLAN=$f timeout 300 python tools/bench_models.py 2>&1 | grep "DCN-v2 CrossNetV2 depth3 B=512" | sed "s/^/PLAN=$f /"; done
