cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_certify_gpu.py tests/test_pyramid_gpu.py tests/test_fuzz_gpu.py tests/test_exact_order_gpu.py tests/test_configs_gpu.py -q -m gpu -x 2>&1 | grep -E "passed|failed|rror" | tail -3
timeout 300 python tools/cert_by_tile.py 2>&1 | grep -E "tile \(|level" | cut -c1-260 | head -60
