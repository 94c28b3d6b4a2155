cd $GRAFT_REPO_ROOT
PYR_TRACE=1 PYR_ONLY=0,0,7 timeout 400 python tools/pyr_profile.py 1024 2>&1 | grep -v amdgpu | cut -c1-250 | head -120
