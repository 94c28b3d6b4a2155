"""Zone shapes of the levels of one pyramid tile (OPT_TRACE bit 1) for the correlate tool's defaults (LoG 1.4 + NCC 11x11).  GPU box only."""
import sys
import numpy as np, torch
sys.path.insert(0, ".")
from visionworkbench_amd import stereo, synth, core
from visionworkbench_amd.core import BBox2i
W = 4096
L, R, _ = synth.stereo_pair(W, W, 129, 1)
Lg, Rg = torch.from_numpy(L).cuda(), torch.from_numpy(R[:, 64:64 + W].copy()).cuda()
ctx = core.default_context(0)
pf, cost, k = (int(a) for a in sys.argv[1:4]) if len(sys.argv) > 3 else (2, 2, 11)
run = lambda: stereo.pyramid_correlate(Lg, Rg, None, None, pf, 1.4 if pf else 0.0, BBox2i.from_corners((-64, -1), (64, 1)), (k, k), cost, consistency_threshold=2, filter_half_kernel=5, max_pyramid_levels=5, bbox=BBox2i(1024, 1024, 1024, 1024))
run(); torch.cuda.synchronize()
ctx.set_option(core.OPT_TRACE, 2)
run(); torch.cuda.synchronize()
