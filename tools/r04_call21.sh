cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
c="2,2,11"; tag=r04g_2_2_11
PYR_ONLY=$c timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_WAIT_INST_ANY -d /tmp/pmcz_$tag -o pmcz -- python tools/pyr_profile.py 1024 > /tmp/pmcz_$tag.log 2>&1
db=$(find /tmp/pmcz_$tag -name "*.db" | head -1)
python tools/rocprof_summary.py "$db" gpurun_out/zones_pmc_$tag.md > /dev/null 2>&1
grep -E "bm_zones_kernel<2, 11, double, true" gpurun_out/zones_pmc_$tag.md | head -30
PYR_ONLY=$c timeout 300 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_WAIT_ANY SQ_INSTS_VMEM_RD SQ_LDS_IDX_ACTIVE SQ_LDS_DATA_FIFO_FULL -d /tmp/pmcz2_$tag -o pmcz2 -- python tools/pyr_profile.py 1024 > /tmp/pmcz2_$tag.log 2>&1
db=$(find /tmp/pmcz2_$tag -name "*.db" | head -1)
python tools/rocprof_summary.py "$db" gpurun_out/zones_pmc2_$tag.md > /dev/null 2>&1
grep -E "bm_zones_kernel<2, 11, double, true" gpurun_out/zones_pmc2_$tag.md | head -30
PYR_LAUNCHES=1 PYR_ONLY=$c timeout 300 python tools/pyr_profile.py 1024 2>&1 | grep -v amdgpu | cut -c1-400
