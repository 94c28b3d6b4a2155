"""One SGM call on RAGGED boxes in isolation (a previous level's disparity image given): the state of the path aggregation inside a pyramid
tile without the pyramid's data flow, so that knock-out builds can be timed.  usage: python tools/probes/time_sgm_ragged.py [side] [reps]
env SGM_PATH_MODE: VWGPU_OPT_SGM_PATH_MODE (bit 6 = one line per wavefront); BAD: share of untrusted coarser pixels (0.002); MGM=1: the MGM passes"""
import os
import sys
import numpy as np, torch
sys.path.insert(0, ".")
from visionworkbench_amd import stereo, synth, core
from visionworkbench_amd.core import BBox2i
W = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
REPS = int(sys.argv[2]) if len(sys.argv) > 2 else 5
BAD = float(os.environ.get("BAD", "0.002"))            # share of pixels without a coarser disparity (they search the whole range)
MGM = os.environ.get("MGM", "0") == "1"
k = 7; hk = k // 2
L, R, truth = synth.stereo_pair(W + 2 * hk, W + 2 * hk, 129, 3)
ctx = core.default_context(0)
if os.environ.get("SGM_PATH_MODE"): ctx.set_option(19, int(os.environ["SGM_PATH_MODE"]))
prev = np.zeros(((W + 1) // 2, (W + 1) // 2, 3), np.int32)
t = truth[hk:hk + W:2, hk:hk + W:2]
prev[..., 0] = t // 2; prev[..., 1] = 0; prev[..., 2] = np.iinfo(np.int32).max
rng = np.random.default_rng(5)
prev[rng.random(prev.shape[:2]) < BAD, 2] = 0
Lg, Rg, prevg = torch.from_numpy(L).cuda(), torch.from_numpy(R).cuda(), torch.from_numpy(prev).cuda()
run = lambda: stereo.calc_disparity_sgm(3, Lg, Rg, BBox2i(hk, hk, W, W), (128, 2), (k, k), prev_disparity=prevg, use_mgm=MGM, ctx=ctx)
run(); torch.cuda.synchronize()
ctx.profile_enable(True)
best = {}
for _ in range(REPS):
    ctx.profile_reset(); run(); torch.cuda.synchronize()
    for n, ms in ctx.profile_read(1 << 12):
        if n.startswith("sgm_"): best[n] = min(best.get(n, 1e9), ms)
print("  ".join("%s %.3f" % (n, ms) for n, ms in sorted(best.items(), key=lambda kv: -kv[1])[:6]))
