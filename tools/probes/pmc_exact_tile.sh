# PMC counters of the exact-order kernels inside one LoG + NCC pyramid tile (tools/pyr_profile.py).  GPU box only.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
export PYR_ONLY=2,2,11 PYR_EXACT_SPLIT=${PYR_EXACT_SPLIT:-3}
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAIT_INST_ANY SQ_INSTS_VMEM_RD -d /tmp/pmcx -o pmcx -- python tools/pyr_profile.py 1024 > /tmp/pmcx.log 2>&1
db=$(find /tmp/pmcx -name "*.db" | head -1)
python tools/rocprof_summary.py "$db" gpurun_out/exact_tile_pmc.md > /dev/null 2>&1
grep -E "bmx_|counter|kernel \|" gpurun_out/exact_tile_pmc.md | head -80
