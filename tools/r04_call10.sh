cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_certify_gpu.py tests/test_fuzz_gpu.py -q -m gpu -x -k "certif or pyramid" 2>&1 | grep -v "^certification" | tail -4
PYR_ONLY=2,2,11 PYR_LAUNCHES=1 timeout 400 python tools/pyr_profile.py 1024 2>&1 | grep -E "^==|bm_zones|launches \(us\)" | cut -c1-500
PYR_ONLY="LoG 1.4 + NCC" timeout 400 python tools/pyr_throughput.py 4 2>&1 | grep -v amdgpu
python - <<'PY' 2>&1 | grep -v amdgpu | tail -5
import sys
sys.path.insert(0, ".")
import torch
from visionworkbench_amd import stereo, synth, core
from visionworkbench_amd.core import BBox2i
L, R, _ = synth.stereo_pair(4096, 4096, 129, 1)
Lg, Rg = torch.from_numpy(L).cuda(), torch.from_numpy(R[:, 64:64 + 4096].copy()).cuda()
ctx = core.default_context(0)
ctx.set_option(core.OPT_TRACE, 4)
stereo.pyramid_correlate(Lg, Rg, None, None, 2, 1.4, BBox2i.from_corners((-64, -1), (64, 1)), (11, 11), 2,
                         consistency_threshold=2, filter_half_kernel=5, max_pyramid_levels=5, bbox=BBox2i(256, 256, 1024, 1024))
torch.cuda.synchronize()
print("certified per mille:", ctx.get_option(core.OPT_CERT_PERMILLE))
PY
