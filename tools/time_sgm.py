"""Device-resident timing of calc_disparity_sgm + per-kernel breakdown.  GPU box only."""
import sys, time, collections
import numpy as np, torch
sys.path.insert(0, ".")
from visionworkbench_amd import stereo, synth, core
from visionworkbench_amd.core import BBox2i
W = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
H = int(sys.argv[2]) if len(sys.argv) > 2 else W
SX = int(sys.argv[3]) if len(sys.argv) > 3 else 128
MGM = len(sys.argv) > 4 and sys.argv[4] == "mgm"
L, R, _ = synth.stereo_pair(W, H, SX + 1, 1)
Lg, Rg = torch.from_numpy(L).cuda(), torch.from_numpy(R).cuda()
ctx = core.default_context(0)
import os
if os.environ.get("SGM_PATH_MODE"): ctx.set_option(19, int(os.environ["SGM_PATH_MODE"]))      # VWGPU_OPT_SGM_PATH_MODE
run = lambda: stereo.calc_disparity_sgm(3, Lg, Rg, BBox2i(0, 0, W, H), (SX, 0), (7, 7), use_mgm=MGM, with_subpixel=True, memory_limit_mb=200000)
run(); torch.cuda.synchronize()
t0 = time.perf_counter(); out = run(); torch.cuda.synchronize(); wall = time.perf_counter() - t0
ctx.profile_enable(True); ctx.profile_reset()
run(); torch.cuda.synchronize()
rec = ctx.profile_read(1 << 12)
ctx.profile_enable(False)
print("calc_disparity_sgm%s %dx%d D=%d: wall %.1f ms = %.1f Mpix/s" % (" (MGM)" if MGM else "", W, H, SX + 1, wall * 1e3, W * H / wall / 1e6))
for n, ms in rec:
    print("  %-18s %.2f ms" % (n, ms))
