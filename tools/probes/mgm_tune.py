"""use_mgm path aggregation: one launch per front vs the four concurrent sweeps (VWGPU_OPT_MGM_SWEEP).  GPU box only."""
import sys
import numpy as np, torch
sys.path.insert(0, ".")
from visionworkbench_amd import stereo, synth, core
from visionworkbench_amd.core import BBox2i
ctx = core.default_context(0)
for W in (1024, 2048):
    L, R, _ = synth.stereo_pair(W, W, 129, 1)
    Lg, Rg = torch.from_numpy(L).cuda(), torch.from_numpy(R).cuda()
    f = lambda: stereo.calc_disparity_sgm(3, Lg, Rg, BBox2i(0, 0, W, W), (128, 0), (7, 7), use_mgm=True, with_subpixel=True, memory_limit_mb=200000, ctx=ctx)
    ref = None
    for opt in (1, 0, 11, 7, 4):
        ctx.set_option(core.OPT_MGM_SWEEP, opt)
        out = f(); torch.cuda.synchronize()
        ctx.profile_enable(True); ctx.profile_reset(); f(); torch.cuda.synchronize(); rec = ctx.profile_read(1 << 14); ctx.profile_enable(False)
        t = sum(m for n, m in rec if n == "sgm_mgm_paths")
        o = out[0].cpu().numpy()
        if ref is None: ref = o
        print("%d^2 x 129, VWGPU_OPT_MGM_SWEEP %2d (1 = one launch per front, 0 = sweeps, n = sweeps with n lines per workgroup): sgm_mgm_paths %.2f ms, identical %s"
              % (W, opt, t, np.array_equal(o, ref)), flush=True)
