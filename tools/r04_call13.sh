cd $GRAFT_REPO_ROOT
for lib in visionworkbench_amd/lib/libvwgpu.so tools/build/libvwgpu_varB.so visionworkbench_amd/lib/libvwgpu.so tools/build/libvwgpu_varB.so; do
  VWGPU_LIBRARY=$PWD/$lib python bench.py --steps 800 --warmup 50 --no-cpu-baseline --no-extra --no-traffic 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print('$lib', round(d['ms_per_step']*1e3,2), d['roofline']['avg_us_per_launch'])"
done
