cd $GRAFT_REPO_ROOT
VWGPU_ZONES16=1 python tools/zones_timeline.py 0 0 7 2>&1 | grep -v amdgpu | tail -12
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
VWGPU_ZONES16=1 VWGPU_LIBRARY=$PWD/tools/build/libvwgpu_stamps.so PYR_ONLY=0,0,7 timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_WAIT_INST_ANY -d /tmp/pmcz16 -o pmcz -- python tools/pyr_profile.py 1024 > /tmp/pmcz16.log 2>&1
db=$(find /tmp/pmcz16 -name "*.db" | head -1)
python tools/rocprof_summary.py "$db" gpurun_out/zones16_pmc.md > /dev/null 2>&1
grep -E "bm_zones_kernel" gpurun_out/zones16_pmc.md | head -24
