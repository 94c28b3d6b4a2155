"""Share of the certified pixels of each 1024^2 tile of the 4096^2 pair (LoG 1.4 + NCC 11x11 through the pyramid) that the fp32 tier
decided / that went on to the float64 tier / to the exact-order kernels (VWGPU_OPT_CERT_PERMILLE, _CERT_F64_PERMILLE).  GPU box only."""
import sys
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
        ctx.set_option(core.OPT_TRACE, 4)
        stereo.pyramid_correlate(Lg, Rg, None, None, 2, 1.4, BBox2i.from_corners((-64, -1), (64, 1)), (11, 11), 2, consistency_threshold=2,
                                 filter_half_kernel=5, max_pyramid_levels=5, bbox=BBox2i(x, y, tile, tile))
        torch.cuda.synchronize()
        sys.stderr.flush()
        print("tile (%4d,%4d): certified %d per mille of the pixels of certified passes, float64 tier %d per mille" %
              (x, y, ctx.get_option(core.OPT_CERT_PERMILLE), ctx.get_option(core.OPT_CERT_F64_PERMILLE)), flush=True)
        ctx.set_option(core.OPT_TRACE, 0)
