cd $GRAFT_REPO_ROOT
for lib in visionworkbench_amd/lib/libvwgpu.so tools/build/libvwgpu_varP.so visionworkbench_amd/lib/libvwgpu.so tools/build/libvwgpu_varP.so; do
  VWGPU_LIBRARY=$PWD/$lib python bench.py --steps 800 --warmup 50 --no-cpu-baseline --no-extra --no-traffic 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print('$lib', round(d['ms_per_step']*1e3,2), d['roofline']['avg_us_per_launch'])"
done
VWGPU_LIBRARY=$PWD/tools/build/libvwgpu_varP.so python -m pytest tests/test_bm_gpu.py -q -m gpu -x -k "sad or config or ties" 2>&1 | tail -2
