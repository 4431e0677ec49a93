#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
timeout 1200 python -m pytest $R/tests/test_deepfm_gpu.py $R/tests/test_deepfm_step_c.py $R/tests/test_trainer.py $R/tests/test_checkpoint.py $R/tests/test_reference_entrypoint.py $R/tests/test_compat_gpu.py $R/tests/test_xdeepfm.py $R/tests/test_autograd.py $R/tests/test_sharded.py -m gpu -x -q 2>&1 | tail -5
