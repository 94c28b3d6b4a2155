"""SSD / NCC matchers at 4096^2 x 129 disparities: register-blocked kernel (bm_corr_u8), per-kernel times from HIP events.  GPU box only."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, ".")
from visionworkbench_amd import stereo, synth, core
from visionworkbench_amd.core import BBox2i
W = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
L, R, _ = synth.stereo_pair(W, W, 129, 1)
Lg, Rg = torch.from_numpy(L).cuda(), torch.from_numpy(R).cuda()
box = BBox2i(0, 0, W, W)
ctx = core.default_context(0)
ctx.set_option(core.OPT_DEFER_EXACTNESS, 1)
for env in ("",):
    for cost, k in [(1, 7), (2, 7), (1, 11), (2, 11), (2, 5)]:
        fn = lambda: stereo.calc_disparity(cost, Lg, Rg, box, (129, 1), (k, k), ctx=ctx)
        a = fn(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5): fn()
        torch.cuda.synchronize(); ms = (time.perf_counter() - t0) / 5 * 1e3
        ctx.profile_reset(); ctx.profile_enable(True); fn(); torch.cuda.synchronize(); ctx.profile_enable(False)
        rec = ctx.profile_read(64)
        print("%s cost=%d k=%d: %.3f ms path=%d  %s" % ("dot " if env else "corr", cost, k, ms, ctx.last_path(), " ".join("%s=%.3f" % (n, m) for n, m in rec)))
