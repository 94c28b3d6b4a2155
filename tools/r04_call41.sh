cd $GRAFT_REPO_ROOT
timeout 300 python tools/zones_ab.py 2>&1 | grep prefilter
