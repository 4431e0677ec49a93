cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -m gpu -x -q -n 4 2>&1 | tail -4
bash tools/runs/r06_small_final.sh 2>&1 | grep -v "^+"
