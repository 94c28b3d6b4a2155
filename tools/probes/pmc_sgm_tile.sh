cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAIT_INST_ANY SQ_INSTS_LDS SQ_ACTIVE_INST_LDS -d /tmp/pmc1 -o pmc1 -- python tools/pyr_sgm_profile.py > /tmp/pmc1.log 2>&1
db=$(find /tmp/pmc1 -name "*.db" | head -1)
python tools/rocprof_summary.py "$db" gpurun_out/sgm_tile_pmc.md > /dev/null 2>&1
grep -E "path_inplace_kernel<8>|path_inplace_kernel<4>" gpurun_out/sgm_tile_pmc.md | head -30
