for v in S8 S16 S24; do echo "== $v"; VWGPU_LIBRARY=$PWD/build_exp/lib$v.so python tools/time_sad_sx.py | grep -E "sx= 64|sx=129"; done
