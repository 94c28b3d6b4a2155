cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_certify_gpu.py tests/test_pyramid_gpu.py tests/test_fuzz_gpu.py tests/test_configs_gpu.py -q -m gpu -x 2>&1 | grep -E "passed|failed|rror" | tail -3
timeout 400 python tools/pyr_profile.py 1024 2>&1 | grep -E "^==" | cut -c1-300
timeout 400 python tools/pyr_throughput.py 4 2>&1 | grep thr
