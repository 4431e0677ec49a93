#!/bin/bash
# ReLU masks as bits between a layer's forward and the dX GEMM of its backward (REC_RELU_BITS=0: the activation as before):
# GEMM + model tests, three interleaved pairs of the bench step
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r06relubits; mkdir -p "$O"; cd "$R"
timeout 1200 python -m pytest tests/test_gemm_gpu.py tests/test_deepfm_gpu.py tests/test_deepfm_step_c.py tests/test_dcn_v2_gpu.py tests/test_models_fullsize_gpu.py -x -q -m gpu 2>&1 | tail -3 | tee "$O/pytest.txt"
for i in 1 2 3; do
  for v in 0 1; do
    echo -n "REC_RELU_BITS=$v  "
    REC_RELU_BITS=$v timeout 600 python bench.py --no-other-configs --no-cpu-baseline 2>> "$O/bench.err" | tail -1 | \
      python -c "import json,sys; d=json.loads(sys.stdin.read()); print('ms_per_step %.4f' % d['ms_per_step'], {k: round(v, 4) for k, v in d['kernels_ms'].items()}, 'loss', d['config'].get('loss'))"
  done
done | tee "$O/ab.txt"
