"""Timeline of the fused SGM sweeps (tools build with -DVWGPU_SWEEP_DEBUG, VWGPU_SWEEP_TRACE=1).  GPU box only."""
import os, sys
import numpy as np, torch
sys.path.insert(0, ".")
from visionworkbench_amd import stereo, synth, core
from visionworkbench_amd.core import BBox2i
ctx = core.default_context(0)
W = H = 2048
L, R, _ = synth.stereo_pair(W, H, 129, 1)
Lg, Rg = torch.from_numpy(L).cuda(), torch.from_numpy(R).cuda()
f = lambda: stereo.calc_disparity_sgm(3, Lg, Rg, BBox2i(0, 0, W, H), (128, 0), (7, 7), with_subpixel=True, memory_limit_mb=200000, ctx=ctx)
ctx.set_option(core.OPT_SGM_SWEEP, int(sys.argv[1]) if len(sys.argv) > 1 else 1)
f(); torch.cuda.synchronize()
os.environ["VWGPU_SWEEP_TRACE"] = "1"
f(); torch.cuda.synchronize()
