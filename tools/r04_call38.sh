cd $GRAFT_REPO_ROOT
for a in 0 0.35 0.6; do echo "== ramp $a"; VWGPU_ZRAMP=$a timeout 300 python tools/zones_ab.py 2>&1 | grep prefilter; done
