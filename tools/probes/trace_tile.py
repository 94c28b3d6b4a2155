import sys, time
import numpy as np, torch
sys.path.insert(0, ".")
from visionworkbench_amd import stereo, synth, core
from visionworkbench_amd.core import BBox2i
W = 4096
L, R, _ = synth.stereo_pair(W, W, 129, 1)
Lg, Rg = torch.from_numpy(L).cuda(), torch.from_numpy(R[:, 64:64 + W].copy()).cuda()
ctx = core.default_context(0)
run = lambda: stereo.pyramid_correlate(Lg, Rg, None, None, 0, 0.0, BBox2i.from_corners((-64, -1), (64, 1)), (7, 7), 0, consistency_threshold=2, filter_half_kernel=5, max_pyramid_levels=5, bbox=BBox2i(1024, 1024, 1024, 1024))
run(); run(); torch.cuda.synchronize()
ctx.set_option(core.OPT_TRACE, 1)
run(); torch.cuda.synchronize()
