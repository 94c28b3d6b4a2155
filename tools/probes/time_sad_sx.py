"""bm_sad_u8 time vs search width on the 4096^2 pair: the intercept is staging + epilogue, the slope the step loop.  GPU box only."""
import sys, time
import numpy as np, torch
sys.path.insert(0, ".")
from visionworkbench_amd import stereo, synth, core
from visionworkbench_amd.core import BBox2i

W = 4096
L, R, _ = synth.stereo_pair(W, W, 129, 1)
Lg, Rg = torch.from_numpy(L).cuda(), torch.from_numpy(R).cuda()
for sx in (4, 8, 16, 32, 64, 128, 129, 256):
    Rc = Rg[:, :W + sx - 1].contiguous() if sx <= 129 else torch.cat([Rg, Rg[:, :sx - 129]], 1).contiguous()
    fn = lambda: stereo.calc_disparity(0, Lg, Rc, BBox2i(0, 0, W, W), (sx, 1), (7, 7))
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20): fn()
    torch.cuda.synchronize()
    us = (time.perf_counter() - t0) / 20 * 1e6
    ctx = core.default_context(0)
    ctx.profile_reset(); ctx.profile_enable(True)
    for _ in range(5): fn()
    torch.cuda.synchronize(); ctx.profile_enable(False)
    per = {}
    for name, ms in ctx.profile_read(256): per.setdefault(name, []).append(ms * 1e3)
    ks = "  ".join("%s %.1f" % (k, sum(v) / len(v)) for k, v in per.items())
    print("sx=%3d: %7.1f us   %.3f us per disparity   path=%d   [%s]" % (sx, us, us / sx, ctx.last_path(), ks))
