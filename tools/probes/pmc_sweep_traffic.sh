# HBM traffic of the fused SGM sweeps (VWGPU_OPT_SGM_SWEEP=1, tools/sweep_pmc.py): separate FETCH_SIZE / WRITE_SIZE passes.  GPU box only.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_$c
  rocprofv3 --kernel-trace --pmc $c -d /tmp/pmc_$c -o pmc_$c -- python tools/sweep_pmc.py > /tmp/pmc_$c.log 2>&1
  db=$(find /tmp/pmc_$c -name "*.db" | head -1)
  python tools/rocprof_summary.py "$db" gpurun_out/sweep_traffic_$c.md > /dev/null 2>&1
  grep -E "sweep_uniform_kernel.*$c|cost_row_kernel.*$c|wta_uniform_kernel.*$c" gpurun_out/sweep_traffic_$c.md
done
