cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_certify_gpu.py tests/test_pyramid_gpu.py tests/test_fuzz_gpu.py tests/test_exact_order_gpu.py -q -m gpu -x 2>&1 | grep -v "^certification" | tail -4
PYR_LAUNCHES=1 timeout 400 python tools/pyr_profile.py 1024 2>&1 | grep -E "^==|bm_zones|launches \(us\)" | cut -c1-420
timeout 400 python tools/pyr_throughput.py 4 2>&1 | grep -v amdgpu
