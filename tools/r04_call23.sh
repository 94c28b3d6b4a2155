cd $GRAFT_REPO_ROOT
for kk in "3 16" "6 16" "8 8" "12 8" "16 6"; do set -- $kk
echo "== capk $1 min $2"
VWGPU_ZCAPK=$1 VWGPU_ZCAPMIN=$2 PYR_LAUNCHES=1 PYR_ONLY=2,2,11 timeout 300 python tools/pyr_profile.py 1024 2>&1 | grep -E "^==|launches \(us\)|bm_zones" | cut -c1-330
VWGPU_ZCAPK=$1 VWGPU_ZCAPMIN=$2 PYR_ONLY=0,0,7 timeout 300 python tools/pyr_profile.py 1024 2>&1 | grep -E "^==|bm_zones" | cut -c1-330
done
