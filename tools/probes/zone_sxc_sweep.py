"""Round 6: the float twin's single-level NCC 11x11 / SSD 7x7 (4096^2 x 129) by VWGPU_OPT_ZONE_SXC (disparities per staged right patch).  GPU box only."""
import sys, time
import numpy as np, torch
sys.path.insert(0, ".")
import visionworkbench_amd as vwa
from visionworkbench_amd import core, stereo, synth
N = 4096
left, right, _ = synth.stereo_pair(N, N, 129, 1)
rng = np.random.default_rng(20260926)
lf = torch.from_numpy((left * np.float32(0.37) + rng.random(left.shape, dtype=np.float32)).astype(np.float32)).cuda()
rf = torch.from_numpy((right * np.float32(0.37) + rng.random(right.shape, dtype=np.float32)).astype(np.float32)).cuda()
ctx = vwa.Context(0)
bb = vwa.BBox2i(0, 0, N, N)
ref = {}
for sxc in [int(a) for a in sys.argv[1:]] or [0, 8, 16, 24, 32, 48, 64, 129]:
    ctx.set_option(core.OPT_ZONE_SXC, sxc)
    for cost, k in ((1, 7), (2, 11)):
        fn = lambda: stereo.calc_disparity(cost, lf, rf, bb, (129, 1), (k, k), ctx=ctx)
        o = fn(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5): o = fn()
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
        if (cost, k) not in ref: ref[(cost, k)] = o.clone()
        print("sxc %3d cost %d %2dx%-2d: %.3f ms  %s" % (sxc, cost, k, k, dt * 1e3, "identical" if torch.equal(o, ref[(cost, k)]) else "DIFFERENT"), flush=True)
