cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
t0=$(date +%s)
timeout 900 python -m pytest tests/test_certify_gpu.py tests/test_pyramid_gpu.py tests/test_fuzz_gpu.py tests/test_exact_order_gpu.py -q -m gpu -x 2>&1 | grep -v "^certification" | tail -15
echo "tests done $(( $(date +%s) - t0 )) s"
PYR_LAUNCHES=1 timeout 400 python tools/pyr_profile.py 1024 2>&1 | grep -v amdgpu > gpurun_out/pyr_profile_r04e.txt; cat gpurun_out/pyr_profile_r04e.txt
timeout 400 python tools/pyr_throughput.py 4 2>&1 | grep -v amdgpu > gpurun_out/pyr_throughput_r04e.txt; cat gpurun_out/pyr_throughput_r04e.txt
echo "total $(( $(date +%s) - t0 )) s"
