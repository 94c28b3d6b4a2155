# rocprofv3 kernel trace of the four-thread pyramid tile loops (tools/pyr_throughput.py): per-kernel totals and how busy the GPU was.
# usage: TAG=r04k bash tools/prof_tile_loop.sh   (GPU box)
TAG=${TAG:-r04k}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for c in "SAD 7x7, integer" "LoG 1.4 + NCC"; do
  tag=$(echo "$c" | tr -dc 'A-Za-z' | cut -c1-6)
  rm -rf /tmp/ktr_$tag
  PYR_ONLY="$c" timeout 300 rocprofv3 --kernel-trace -d /tmp/ktr_$tag -o ktr -- python tools/pyr_throughput.py 4 > /tmp/ktr_$tag.log 2>&1
  db=$(find /tmp/ktr_$tag -name "*.db" | head -1)
  { echo "## $c — four tile threads, 96 tiles of the 4096^2 pair (rocprofv3 --kernel-trace; the tracer slows the host side)"; grep thr /tmp/ktr_$tag.log; python tools/trace_overlap.py "$db" 0.3; echo; } >> gpurun_out/${TAG}_tile_loop_kernels.md
done
cat gpurun_out/${TAG}_tile_loop_kernels.md
