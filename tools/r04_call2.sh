cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
t0=$(date +%s)
timeout 300 python tools/time_strip_step.py > gpurun_out/strip_step_r04b.txt 2>&1; cat gpurun_out/strip_step_r04b.txt | grep -v amdgpu
timeout 200 python -m pytest tests/test_bm_gpu.py tests/test_configs_gpu.py -q -m gpu -x -k "sad or config1 or config2 or deferred or trim or ties" 2>&1 | tail -3
timeout 400 python tools/pyr_profile.py 1024 > gpurun_out/pyr_profile_r04b.txt 2>&1; grep -v amdgpu gpurun_out/pyr_profile_r04b.txt
PYR_ONLY="" timeout 400 python tools/pyr_throughput.py 4 8 > gpurun_out/pyr_throughput_r04b.txt 2>&1; grep -v amdgpu gpurun_out/pyr_throughput_r04b.txt
# PMC: what bounds bm_zones on an integer NCC tile and on a SAD tile
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for c in "0,2,11" "0,0,7"; do
  tag=$(echo $c | tr , _)
  PYR_ONLY=$c timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_WAIT_INST_ANY -d /tmp/pmcz_$tag -o pmcz -- python tools/pyr_profile.py 1024 > /tmp/pmcz_$tag.log 2>&1
  db=$(find /tmp/pmcz_$tag -name "*.db" | head -1)
  python tools/rocprof_summary.py "$db" gpurun_out/zones_pmc_$tag.md > /dev/null 2>&1
  grep -E "bm_zones|zone_prec" gpurun_out/zones_pmc_$tag.md | head -30
  PYR_ONLY=$c timeout 300 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_WAIT_ANY SQ_INSTS_VMEM_RD SQ_LDS_IDX_ACTIVE SQ_LDS_DATA_FIFO_FULL -d /tmp/pmcz2_$tag -o pmcz2 -- python tools/pyr_profile.py 1024 > /tmp/pmcz2_$tag.log 2>&1
  db=$(find /tmp/pmcz2_$tag -name "*.db" | head -1)
  python tools/rocprof_summary.py "$db" gpurun_out/zones_pmc2_$tag.md > /dev/null 2>&1
  grep -E "bm_zones" gpurun_out/zones_pmc2_$tag.md | head -30
done
echo "total $(( $(date +%s) - t0 )) s"
