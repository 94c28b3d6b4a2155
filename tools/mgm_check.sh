cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 600 python -m pytest tests -x -q -m gpu -k "mgm or torch_device_entry or pyramid_sgm or stereo_surface or cpp" > gpurun_out/mgm_tests.log 2>&1; echo "rc=$?" >> gpurun_out/mgm_tests.log
grep -E "passed|failed|rc=|Error|error" gpurun_out/mgm_tests.log | tail -8
timeout 200 python tools/time_sgm.py 1024 1024 128 mgm 2>&1 | grep -v "^$" | tail -12
timeout 200 python tools/time_sgm.py 1024 1024 128 2>&1 | tail -12
