cd $GRAFT_REPO_ROOT
echo "== product (4 residents, two planes)"; timeout 300 python tools/zones_ab.py 2>&1 | grep "cost 2"
echo "== one plane, 4 residents by registers"; VWGPU_ZONEPLANE=1 timeout 300 python tools/zones_ab.py 2>&1 | grep "cost 2"
echo "== one plane, 96 registers (5 residents)"; VWGPU_ZONEPLANE=1 VWGPU_LIBRARY=$PWD/tools/build/libvwgpu_zlb5.so timeout 300 python tools/zones_ab.py 2>&1 | grep "cost 2"
echo "== two planes, 96 registers"; VWGPU_LIBRARY=$PWD/tools/build/libvwgpu_zlb5.so timeout 300 python tools/zones_ab.py 2>&1 | grep "cost 2"
