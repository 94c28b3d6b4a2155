"""Per-call times of the SGM stages inside one pyramid tile (the 12 sgm_paths calls: six levels, left and right).  usage: python tools/probes/pyr_sgm_levels.py [algorithm]"""
import sys
import numpy as np, torch
sys.path.insert(0, ".")
from visionworkbench_amd import stereo, synth, core
from visionworkbench_amd.core import BBox2i
W = 4096
ALG = int(sys.argv[1]) if len(sys.argv) > 1 else 1
L, R, _ = synth.stereo_pair(W, W, 129, 1)
Lg, Rg = torch.from_numpy(L).cuda(), torch.from_numpy(R[:, 64:64 + W].copy()).cuda()
ctx = core.default_context(0)
import os
if os.environ.get("SGM_PATH_MODE"): ctx.set_option(19, int(os.environ["SGM_PATH_MODE"]))
run = lambda: stereo.pyramid_correlate(Lg, Rg, None, None, 0, 0.0, BBox2i.from_corners((-64, -1), (64, 1)), (7, 7), 3, consistency_threshold=2,
                                       filter_half_kernel=5, max_pyramid_levels=5, algorithm=ALG, bbox=BBox2i(1024, 1024, 1024, 1024))
run(); torch.cuda.synchronize()
ctx.profile_enable(True); ctx.profile_reset(); run(); torch.cuda.synchronize()
for n, ms in ctx.profile_read(1 << 14):
    if n.startswith("sgm_") and ms > 0.02: print("%-20s %.3f ms" % (n, ms))
