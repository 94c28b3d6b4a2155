cd $GRAFT_REPO_ROOT
timeout 400 python tools/pyr_throughput.py 4 2>&1 | grep -v amdgpu
