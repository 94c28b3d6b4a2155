"""sgm_paths time of the fused sweeps for several rows-per-workgroup settings.  GPU box only."""
import sys, time
import numpy as np, torch
sys.path.insert(0, ".")
from visionworkbench_amd import stereo, synth, core
from visionworkbench_amd.core import BBox2i
ctx = core.default_context(0)
W = H = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
L, R, _ = synth.stereo_pair(W, H, 129, 1)
Lg, Rg = torch.from_numpy(L).cuda(), torch.from_numpy(R).cuda()
f = lambda: stereo.calc_disparity_sgm(3, Lg, Rg, BBox2i(0, 0, W, H), (128, 0), (7, 7), with_subpixel=True, memory_limit_mb=200000, ctx=ctx)
for nw in [0, 1, 13, 7, 4]:
    ctx.set_option(core.OPT_SGM_SWEEP, nw)
    f(); torch.cuda.synchronize()
    ctx.profile_enable(True); ctx.profile_reset(); f(); f(); torch.cuda.synchronize(); rec = ctx.profile_read(4096); ctx.profile_enable(False)
    t = [m for n, m in rec if n == "sgm_paths"]
    print("VWGPU_OPT_SGM_SWEEP %2d (0 = per-direction launches, 1 = sweeps, n = sweeps with n rows per workgroup): sgm_paths %s ms" % (nw, " ".join("%.2f" % x for x in t)), flush=True)
