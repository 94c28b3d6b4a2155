cd $GRAFT_REPO_ROOT
PYR_LAUNCHES=1 timeout 400 python tools/pyr_profile.py 1024 2>&1 | grep -E "^==|launches \(us\)" | cut -c1-300
