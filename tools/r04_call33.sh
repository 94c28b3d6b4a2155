cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_pyramid_gpu.py -q -m gpu -x 2>&1 | grep -v "^certification" | tail -30
