"""Per-kernel times of parabola_subpixel 11x11 on the 4096^2 NCC result (config 3b).  GPU box only."""
import sys, torch, numpy as np
sys.path.insert(0, ".")
from visionworkbench_amd import stereo, synth, core
from visionworkbench_amd.core import BBox2i
W = 4096
L, R, _ = synth.stereo_pair(W, W, 129, 1)
Lg, Rg = torch.from_numpy(L).cuda(), torch.from_numpy(R).cuda()
ctx = core.default_context(0)
d = stereo.calc_disparity(2, Lg, Rg, BBox2i(0, 0, W, W), (129, 1), (11, 11), ctx=ctx)
disp = torch.zeros((W, W, 3), dtype=torch.float32, device="cuda")
disp[5:5 + W - 10, 5:5 + W - 10, :2] = d[..., :2].float(); disp[5:5 + W - 10, 5:5 + W - 10, 2] = (d[..., 2] != 0).float()
f = lambda: stereo.parabola_subpixel(disp, Lg, Rg, 0, 0.0, (11, 11), ctx=ctx)
f(); torch.cuda.synchronize()
for _ in range(2):
    ctx.profile_reset(); ctx.profile_enable(True); f(); torch.cuda.synchronize(); ctx.profile_enable(False)
    print(" ".join("%s=%.3f" % (n, m) for n, m in ctx.profile_read(64)))
