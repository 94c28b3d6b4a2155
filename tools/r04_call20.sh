cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
PYR_TRACE=2 PYR_ONLY=2,2,11 PYR_LAUNCHES=1 timeout 400 python tools/pyr_profile.py 1024 2>&1 | grep -v amdgpu > gpurun_out/pyr_trace_r04g.txt
tail -150 gpurun_out/pyr_trace_r04g.txt | cut -c1-300
