cd $GRAFT_REPO_ROOT
PYR_TRACE_PLAIN=1 PYR_ONLY=0,0,7 timeout 400 python tools/pyr_profile.py 1024 2>&1 | grep -v amdgpu | cut -c1-250 | grep -A30 "level 1 zones of" | head -42
