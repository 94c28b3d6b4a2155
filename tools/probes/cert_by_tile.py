"""Which tiles of the 4096^2 pair send zones to the exact-order kernels?  Per tile: certification line (TRACE bit 2), kernel time of
the tile kernels and of the exact-order kernels.  GPU box only."""
import sys, collections
import numpy as np, torch
sys.path.insert(0, ".")
from visionworkbench_amd import stereo, synth, core
from visionworkbench_amd.core import BBox2i
W, tile = 4096, 1024
L, R, _ = synth.stereo_pair(W, W, 129, 1)
Lg, Rg = torch.from_numpy(L).cuda(), torch.from_numpy(R[:, 64:64 + W].copy()).cuda()
ctx = core.default_context(0)
for y in range(0, W, tile):
    for x in range(0, W, tile):
        run = lambda: stereo.pyramid_correlate(Lg, Rg, None, None, 2, 1.4, BBox2i.from_corners((-64, -1), (64, 1)), (11, 11), 2,
                                               consistency_threshold=2, filter_half_kernel=5, max_pyramid_levels=5, bbox=BBox2i(x, y, tile, tile))
        run(); torch.cuda.synchronize()
        ctx.set_option(core.OPT_TRACE, 4)
        ctx.profile_enable(True); ctx.profile_reset(); run(); torch.cuda.synchronize()
        rec = ctx.profile_read(1 << 16); ctx.profile_enable(False)
        ctx.set_option(core.OPT_TRACE, 0)
        bx = sum(t for n, t in rec if n.startswith("bmx_")); z = sum(t for n, t in rec if n.startswith("bm_zones")); tot = sum(t for _, t in rec)
        sys.stderr.flush()
        print("tile (%4d,%4d): kernels %.2f ms, zone matcher %.2f, exact-order kernels %.2f ms in %d launches" % (x, y, tot, z, bx, sum(1 for n, _ in rec if n.startswith("bmx_"))), flush=True)
