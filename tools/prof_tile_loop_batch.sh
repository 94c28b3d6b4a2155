# rocprofv3 kernel trace of the tile loop fed in GROUPS (tools/tile_loop_batch.py, vwgpu_pyramid_correlate_batch_dev): per-kernel totals, how busy
# the GPU was, dispatches per tile.  usage: TAG=r05 COMBO=4x4 SIZE=4096 bash tools/prof_tile_loop_batch.sh   (GPU box)
TAG=${TAG:-r05}; COMBO=${COMBO:-4x4}; SIZE=${SIZE:-4096}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for only in SAD LoG; do
  rm -rf /tmp/ktr_batch
  TLB_ONLY=$only TLB_REPS=12 timeout 600 rocprofv3 --kernel-trace -d /tmp/ktr_batch -o ktr -- python tools/tile_loop_batch.py $SIZE $COMBO > /tmp/ktr_batch.log 2>&1
  db=$(find /tmp/ktr_batch -name "*.db" | head -1)
  { echo "## tile loop in groups ($COMBO = tile threads x tiles per group), ${SIZE}^2 pair, $only: 13 passes + one with the engine's events (rocprofv3 --kernel-trace; the tracer slows the host side)"; grep "threads" /tmp/ktr_batch.log; python tools/trace_overlap.py "$db" 0.3; echo; } >> gpurun_out/${TAG}_tile_loop_batch_kernels.md
done
cat gpurun_out/${TAG}_tile_loop_batch_kernels.md
