#!/bin/bash
# the weight-gradient kernel's LDS stores under the NEXT tile's MFMAs (REC_X3_DW_PIPE=1, default) against a burst of six behind
# each converted column (=0): GEMM + DeepFM tests, three interleaved pairs of the bench step
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r06dwpipe; mkdir -p "$O"; cd "$R"
timeout 900 python -m pytest tests/test_gemm_gpu.py tests/test_deepfm_gpu.py tests/test_kernel_resources.py -x -q 2>&1 | tail -3 | tee "$O/pytest.txt"
for i in 1 2 3; do
  for v in 0 1; do
    echo -n "REC_X3_DW_PIPE=$v  "
    REC_X3_DW_PIPE=$v timeout 600 python bench.py --no-other-configs --no-cpu-baseline 2>> "$O/bench.err" | tail -1 | \
      python -c "import json,sys; d=json.loads(sys.stdin.read()); print('ms_per_step %.4f' % d['ms_per_step'], {k: round(v, 4) for k, v in d['kernels_ms'].items()}, 'loss', d['config'].get('loss'))"
  done
done | tee "$O/ab.txt"
