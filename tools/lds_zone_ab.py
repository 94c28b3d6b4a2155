"""A/B of the LDS-resident small-zone kernel (OPT_EXACT_LDS 0 = off vs 1 = on) on pyramid tiles.  GPU box only."""
import sys, time, collections
import numpy as np, torch
sys.path.insert(0, ".")
from visionworkbench_amd import stereo, synth, core
from visionworkbench_amd.core import BBox2i
W = 4096
L, R, _ = synth.stereo_pair(W, W, 129, 1)
Lg, Rg = torch.from_numpy(L).cuda(), torch.from_numpy(R[:, 64:64 + W].copy()).cuda()
ctx = core.default_context(0)
for pf, cost, k in [(2, 2, 11), (2, 0, 7), (0, 1, 7)]:
    for opt in (0, 1):
        ctx.set_option(core.OPT_EXACT_LDS, opt)
        run = lambda: stereo.pyramid_correlate(Lg, Rg, None, None, pf, 1.4 if pf else 0.0, BBox2i.from_corners((-64, -1), (64, 1)), (k, k), cost, consistency_threshold=2, filter_half_kernel=5, max_pyramid_levels=5, bbox=BBox2i(256, 256, 1024, 1024))
        out = run(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5): run()
        torch.cuda.synchronize(); wall = (time.perf_counter() - t0) / 5
        ctx.profile_enable(True); ctx.profile_reset(); run(); torch.cuda.synchronize(); rec = ctx.profile_read(1 << 16); ctx.profile_enable(False)
        agg = collections.OrderedDict()
        for n, ms in rec:
            a = agg.setdefault(n, [0, 0.0]); a[0] += 1; a[1] += ms
        print("prefilter %d cost %d k %d  EXACT_LDS=%d: wall %.2f ms kernels %.2f ms | %s" % (pf, cost, k, opt, wall * 1e3, sum(v[1] for v in agg.values()),
              " ".join("%s=%d/%.2f" % (n, c, ms) for n, (c, ms) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:7])), flush=True)
