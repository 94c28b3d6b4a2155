cd $GRAFT_REPO_ROOT
PYR_ONLY=2,2,11 PYR_LAUNCHES=all timeout 400 python tools/pyr_profile.py 1024 2>&1 | grep -v amdgpu | cut -c1-2600
