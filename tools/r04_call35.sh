cd $GRAFT_REPO_ROOT
PYR_ONLY=0,0,7 PYR_LAUNCHES=all timeout 400 python tools/pyr_profile.py 1024 2>&1 | grep -E "^==|zone_lr_need" | cut -c1-200
PYR_ONLY=2,2,11 PYR_LAUNCHES=all timeout 400 python tools/pyr_profile.py 1024 2>&1 | grep -E "^==|zone_lr_need|launches" | cut -c1-1800
