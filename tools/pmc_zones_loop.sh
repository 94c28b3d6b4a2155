# PMC counters of the kernels of the LoG + NCC tile loop fed in groups (tools/tile_loop_batch.py 4096 1x16: one tile thread, groups of 16, so that
# the serialised dispatches of a counter pass are the launches the loop really makes).  Two passes (SQ activity; LDS).  GPU box only.
# usage: TAG=r05 bash tools/pmc_zones_loop.sh
TAG=${TAG:-r05}
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
out=gpurun_out/${TAG}_zones_loop_pmc.md; : > $out
pass() {  # name, counters...
  name=$1; shift
  rm -rf /tmp/pmcz
  TLB_ONLY=LoG TLB_REPS=1 timeout 600 rocprofv3 --kernel-trace --pmc "$@" -d /tmp/pmcz -o pmcz -- python tools/tile_loop_batch.py 4096 1x16 > /tmp/pmcz.log 2>&1
  db=$(find /tmp/pmcz -name "*.db" | head -1)
  python tools/rocprof_summary.py "$db" /tmp/pmcz.md > /dev/null 2>&1 || { echo "summary $name failed" >> $out; return; }
  { echo "## LoG + NCC tile loop, groups of 16, one tile thread — counter pass: $name ($*)"; grep -E "bm_zones_kernel<2, 11, double, true, 32, true, true>|zone_precision_sq|zones_merge_kernel<2, true|rm_outliers|sepconv|counter \||kernel \| calls" /tmp/pmcz.md | head -80; echo; } >> $out
}
pass sq SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY
pass lds SQ_WAVE_CYCLES SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU
cat $out | cut -c1-260
