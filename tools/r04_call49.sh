cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_certify_gpu.py -q -m gpu -x 2>&1 | grep -v "^certification" | tail -8
