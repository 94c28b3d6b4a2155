"""Exact-order kernels on single rasters shaped like the large zones of a pyramid level (bmx_col / bmx_row times).  GPU box only."""
import sys, time
import numpy as np, torch
sys.path.insert(0, ".")
import visionworkbench_amd as vwa
from visionworkbench_amd import stereo, core, filters, synth
ctx = core.default_context(0)
rng = np.random.default_rng(1)
for (w, h, sx, sy, k, cost) in [(256, 256, 68, 3, 11, 2), (256, 256, 68, 3, 11, 0), (256, 256, 20, 3, 11, 2), (128, 128, 60, 3, 11, 2), (256, 32, 68, 3, 11, 2), (32, 256, 68, 3, 11, 2),
                                (1024, 1024, 9, 3, 11, 2)]:
    left = rng.random((h + k - 1, w + k - 1)).astype(np.float32) * 3.7
    right = rng.random((h + k - 1 + sy - 1, w + k - 1 + sx - 1)).astype(np.float32) * 3.7
    lt, rt = torch.from_numpy(left).cuda(), torch.from_numpy(right).cuda()
    ctx.force_path(core.PATH_EXACT_ORDER)
    f = lambda: stereo.calc_disparity(cost, lt, rt, vwa.bounding_box(left), (sx, sy), (k, k), ctx=ctx)
    f(); torch.cuda.synchronize()
    ctx.profile_enable(True); ctx.profile_reset(); f(); torch.cuda.synchronize(); rec = ctx.profile_read(4096); ctx.profile_enable(False)
    ctx.force_path(core.PATH_NONE)
    print("%4d x %4d outputs, search %2d x %d, cost %d: %s" % (w, h, sx, sy, cost, "  ".join("%s %.3f ms" % (n, m) for n, m in rec)), flush=True)
