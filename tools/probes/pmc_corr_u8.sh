# PMC counters of the packed NCC / SSD matcher (bm_corr_u8_kernel) on config 3a (4096^2, 11 x 11 NCC, 129 x 1).  GPU box only.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
cat > /tmp/c3a.py <<'PY'
import sys, torch
sys.path.insert(0, ".")
from visionworkbench_amd import stereo, synth, core
from visionworkbench_amd.core import BBox2i
L, R, _ = synth.stereo_pair(4096, 4096, 129, 1)
Lg, Rg = torch.from_numpy(L).cuda(), torch.from_numpy(R).cuda()
for _ in range(3):
    stereo.calc_disparity(2, Lg, Rg, BBox2i(0, 0, 4096, 4096), (129, 1), (11, 11))
torch.cuda.synchronize()
PY
out=gpurun_out/pmc_corr_u8.md; : > $out
pass() { rm -rf /tmp/pmcc; timeout 600 rocprofv3 --kernel-trace --pmc "$@" -d /tmp/pmcc -o p -- python /tmp/c3a.py > /tmp/pmcc.log 2>&1; db=$(find /tmp/pmcc -name "*.db" | head -1); python tools/rocprof_summary.py "$db" /tmp/pmcc.md > /dev/null 2>&1; echo "## $*" >> $out; grep -E "bm_corr_u8|ncc_full" /tmp/pmcc.md >> $out; }
pass SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY
pass SQ_WAVE_CYCLES SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU SQ_INST_CYCLES_VMEM
cat $out | cut -c1-200
