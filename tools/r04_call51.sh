cd $GRAFT_REPO_ROOT
timeout 200 python tools/zones_timeline.py 0 0 7 2>&1 | grep -v amdgpu | cut -c1-330
