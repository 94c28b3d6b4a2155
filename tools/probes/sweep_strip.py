"""Per-direction launches vs fused sweeps on config-4 strip shapes.  GPU box only."""
import sys, time
import numpy as np, torch
sys.path.insert(0, ".")
from visionworkbench_amd import stereo, synth, core
from visionworkbench_amd.core import BBox2i
ctx = core.default_context(0)
for (W, H) in [(2048, 2048), (4096, 1024), (4096, 4096), (8192, 1024), (16384, 2112)]:
    L, R, _ = synth.stereo_pair(W, H, 129, 1)
    Lg, Rg = torch.from_numpy(L).cuda(), torch.from_numpy(R).cuda()
    f = lambda: stereo.calc_disparity_sgm(3, Lg, Rg, BBox2i(0, 0, W, H), (128, 0), (7, 7), with_subpixel=True, memory_limit_mb=200000, ctx=ctx)
    res = {}
    for nw in (0, 1):
        ctx.set_option(core.OPT_SGM_SWEEP, nw)
        out = f(); torch.cuda.synchronize()
        ctx.profile_enable(True); ctx.profile_reset(); f(); torch.cuda.synchronize(); rec = ctx.profile_read(4096); ctx.profile_enable(False)
        res[nw] = (sum(m for n, m in rec if n == "sgm_paths"), sum(m for n, m in rec), out[0].cpu().numpy())
    print("%5d x %5d: sgm_paths per-direction %.2f ms, sweeps %.2f ms; whole call %.2f / %.2f ms; identical %s" % (
        W, H, res[0][0], res[1][0], res[0][1], res[1][1], np.array_equal(res[0][2], res[1][2])), flush=True)
    del Lg, Rg; torch.cuda.empty_cache()
