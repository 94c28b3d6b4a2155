# MGM on the GPU box: the MGM tests, the randomised long forms with the MGM flag, timings.  usage: bash tools/mgm_check.sh
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 900 python -m pytest tests -x -q -m gpu -k "mgm or torch_device_entry or fuzz_sgm or fuzz_pyramid_sgm or cpp" > gpurun_out/mgm_tests.log 2>&1; echo "rc=$?" >> gpurun_out/mgm_tests.log
grep -E "passed|failed|rc=|Error|error" gpurun_out/mgm_tests.log | tail -8
timeout 600 python tools/fuzz_sgm_vs_oracle.py 1500 77 mgm 2>&1 | tail -3
timeout 200 python tools/time_sgm.py 1024 1024 128 mgm 2>&1 | grep -v "^$\|amdgpu.ids" | tail -9
timeout 200 python tools/time_sgm.py 2048 2048 128 mgm 2>&1 | grep -v "^$\|amdgpu.ids" | tail -9
