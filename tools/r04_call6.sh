cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for cfg in "0 0" "1 0" "2 0" "1 16" "1 8" "2 8"; do
  set -- $cfg
  echo "#### ZONE_TILES=$1 ZONE_SXC=$2"
  for only in "0,0,7" "0,2,11" "2,2,11"; do
    PYR_ONLY=$only PYR_ZONE_TILES=$1 PYR_ZONE_SXC=$2 PYR_LAUNCHES=1 timeout 300 python tools/pyr_profile.py 1024 2>&1 | grep -E "^==|bm_zones|launches \(us\)" | cut -c1-400
  done
done
