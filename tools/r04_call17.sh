cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_certify_gpu.py tests/test_pyramid_gpu.py tests/test_fuzz_gpu.py -q -m gpu -x 2>&1 | grep -v "^certification" | tail -3
PYR_LAUNCHES=1 timeout 400 python tools/pyr_profile.py 1024 2>&1 | grep -E "^==|bm_zones" | cut -c1-300
echo "### 16-tiles (tools build)"
VWGPU_ZONES16=1 VWGPU_LIBRARY=$PWD/tools/build/libvwgpu_stamps.so PYR_LAUNCHES=1 timeout 400 python tools/pyr_profile.py 1024 2>&1 | grep -E "^==|bm_zones" | cut -c1-300
timeout 400 python tools/pyr_throughput.py 4 2>&1 | grep -v amdgpu
