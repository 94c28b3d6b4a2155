# needs: make -C visionworkbench_amd/csrc ringdbg
for d in 0 1 2 4 6 7 8 9 15; do
  echo "dbg $d: $(VWGPU_LIBRARY=$PWD/tools/libexp/libvwgpu_ringdbg.so VWGPU_RING_DBG=$d python tools/time_sgm.py 2048 2054 128 2>&1 | grep -E 'sgm_paths' )"
done
