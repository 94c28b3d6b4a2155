cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
t0=$(date +%s)
timeout 600 python -m pytest tests/test_certify_gpu.py -q -m gpu -x 2>&1 | grep -v "^certification" | tail -5
PYR_LAUNCHES=1 timeout 400 python tools/pyr_profile.py 1024 2>&1 | grep -v amdgpu > gpurun_out/pyr_profile_r04d.txt; cat gpurun_out/pyr_profile_r04d.txt
timeout 400 python tools/pyr_throughput.py 4 2>&1 | grep -v amdgpu > gpurun_out/pyr_throughput_r04d.txt; cat gpurun_out/pyr_throughput_r04d.txt
python - <<'PY' 2>&1 | grep -v amdgpu | head -80
import sys
sys.path.insert(0, ".")
import torch
from visionworkbench_amd import stereo, synth, core
from visionworkbench_amd.core import BBox2i
L, R, _ = synth.stereo_pair(4096, 4096, 129, 1)
Lg, Rg = torch.from_numpy(L).cuda(), torch.from_numpy(R[:, 64:64 + 4096].copy()).cuda()
ctx = core.default_context(0)
ctx.set_option(core.OPT_TRACE, 2 | 4)
for pf, cost, k in ((0, 0, 7), (2, 2, 11)):
    print("==== prefilter %d cost %d" % (pf, cost), flush=True)
    stereo.pyramid_correlate(Lg, Rg, None, None, pf, 1.4 if pf else 0.0, BBox2i.from_corners((-64, -1), (64, 1)), (k, k), cost,
                             consistency_threshold=2, filter_half_kernel=5, max_pyramid_levels=5, bbox=BBox2i(256, 256, 1024, 1024))
    torch.cuda.synchronize()
PY
echo "total $(( $(date +%s) - t0 )) s"
