# PMC counters of the exact-order kernels on a LoG + NCC 11x11 pyramid tile.  GPU box only.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
cat > /tmp/one_tile.py <<PY
import sys
import numpy as np, torch
sys.path.insert(0, ".")
from visionworkbench_amd import stereo, synth
from visionworkbench_amd.core import BBox2i
L, R, _ = synth.stereo_pair(4096, 4096, 129, 1)
Lg, Rg = torch.from_numpy(L).cuda(), torch.from_numpy(R[:, 64:64 + 4096].copy()).cuda()
for _ in range(3):
    stereo.pyramid_correlate(Lg, Rg, None, None, 2, 1.4, BBox2i.from_corners((-64, -1), (64, 1)), (11, 11), 2, consistency_threshold=2,
                             filter_half_kernel=5, max_pyramid_levels=5, bbox=BBox2i(256, 256, 1024, 1024))
torch.cuda.synchronize()
PY
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAIT_INST_ANY SQ_INSTS_LDS SQ_INSTS_VMEM_RD -d /tmp/pmcx -o pmcx -- python /tmp/one_tile.py > /tmp/pmcx.log 2>&1
db=$(find /tmp/pmcx -name "*.db" | head -1)
python tools/rocprof_summary.py "$db" gpurun_out/exact_tile_pmc.md > /dev/null 2>&1
grep -E "bmx_row_kernel<2, 8>|bmx_col_kernel<2>" gpurun_out/exact_tile_pmc.md | head -24
