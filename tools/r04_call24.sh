cd $GRAFT_REPO_ROOT
timeout 300 python tools/zones_knockout.py 2 2 11 2>&1 | grep -v amdgpu | cut -c1-300
