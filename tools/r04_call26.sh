cd $GRAFT_REPO_ROOT
PYR_TRACE=2 PYR_ONLY=0,0,7 PYR_LAUNCHES=all timeout 400 python tools/pyr_profile.py 1024 2>&1 | grep -v amdgpu | grep -v "zones  " | cut -c1-1800
