"""A/B timing of the level-0 matcher launches of a pyramid tile (dev aid): median over N runs of the last two bm_zones launches + merges."""
import sys, os, collections, statistics
import numpy as np, torch
sys.path.insert(0, ".")
from visionworkbench_amd import stereo, synth, core
from visionworkbench_amd.core import BBox2i
L, R, _ = synth.stereo_pair(4096, 4096, 129, 1)
Lg, Rg = torch.from_numpy(L).cuda(), torch.from_numpy(R[:, 64:64 + 4096].copy()).cuda()
ctx = core.default_context(0)
for pf, cost, k in [(2, 2, 11), (0, 2, 11), (0, 0, 7)]:
    for (x, y) in [(256, 256), (2048, 1024)]:
        run = lambda: stereo.pyramid_correlate(Lg, Rg, None, None, pf, 1.4 if pf else 0.0, BBox2i.from_corners((-64, -1), (64, 1)), (k, k), cost,
                                               consistency_threshold=2, filter_half_kernel=5, max_pyramid_levels=5, bbox=BBox2i(x, y, 1024, 1024))
        run(); torch.cuda.synchronize()
        zs, ms, tot = [], [], []
        for _ in range(7):
            ctx.profile_enable(True); ctx.profile_reset(); run(); torch.cuda.synchronize()
            rec = ctx.profile_read(1 << 16); ctx.profile_enable(False)
            z = [t * 1e3 for n, t in rec if n == "bm_zones"]; m = [t * 1e3 for n, t in rec if n == "bm_zones_merge"]
            zs.append(sum(z[-2:])); ms.append(sum(m[-2:])); tot.append(sum(t for _, t in rec) * 1e3)
        print("prefilter %d cost %d tile (%d,%d): level-0 matchers %.0f us + merges %.0f us; all kernels %.0f us" %
              (pf, cost, x, y, statistics.median(zs), statistics.median(ms), statistics.median(tot)))
