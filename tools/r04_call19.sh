cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
PYR_LAUNCHES=1 timeout 400 python tools/pyr_profile.py 1024 2>&1 | grep -v amdgpu > gpurun_out/pyr_profile_r04g.txt
cut -c1-400 gpurun_out/pyr_profile_r04g.txt
