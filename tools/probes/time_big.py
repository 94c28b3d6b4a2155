"""Packed SAD matcher on larger pairs (config-4 / config-5 sized strips).  GPU box only."""
import sys, time, numpy as np, torch
sys.path.insert(0, ".")
from visionworkbench_amd import stereo, synth, core
from visionworkbench_amd.core import BBox2i
ctx = core.default_context(0)
for (W, H) in [(4096, 4096), (16384, 2048), (16384, 8192)]:
    L, R, _ = synth.stereo_pair(W, H, 129, 1)
    Lg, Rg = torch.from_numpy(L).cuda(), torch.from_numpy(R).cuda()
    fn = lambda: stereo.calc_disparity(0, Lg, Rg, BBox2i(0, 0, W, H), (129, 1), (7, 7))
    for _ in range(3): fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20): fn()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 20 * 1e3
    ctx.profile_reset(); ctx.profile_enable(True)
    for _ in range(3): fn()
    torch.cuda.synchronize(); ctx.profile_enable(False)
    per = {}
    for name, t in ctx.profile_read(64): per.setdefault(name, []).append(t * 1e3)
    print("%dx%d: %.3f ms  %.1f Gpix/s  path %d  %s" % (W, H, ms, (W - 6) * (H - 6) / ms / 1e6, ctx.last_path(),
          "  ".join("%s %.0f us" % (k, sum(v) / len(v)) for k, v in per.items())))
    del Lg, Rg
