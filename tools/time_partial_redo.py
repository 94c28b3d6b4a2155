import sys, time, numpy as np, torch
sys.path.insert(0, ".")
import visionworkbench_amd as vwa
from visionworkbench_amd import core, stereo, synth
N = 4096
left, right, _ = synth.stereo_pair(N, N, 129, 1)
rng = np.random.default_rng(1)
lf = (left * np.float32(0.37) + rng.random(left.shape, dtype=np.float32)).astype(np.float32)
rf = (right * np.float32(0.37) + rng.random(right.shape, dtype=np.float32)).astype(np.float32)
lf[2000:2040, 1000:1400] = 3.3; rf[2000:2040, 1000:1600] = 3.3          # one flat patch: exact ties
ctx = vwa.Context(0)
lt, rt = torch.from_numpy(lf).cuda(), torch.from_numpy(rf).cuda()
names = {v: k for k, v in vars(core).items() if k.startswith("PATH_")}
for cost, k in ((1, 7), (2, 11)):
    for cert in (1, 0):
        ctx.set_option(core.OPT_CERTIFY, cert)
        fn = lambda: stereo.calc_disparity(cost, lt, rt, vwa.BBox2i(0, 0, N, N), (129, 1), (k, k), ctx=ctx)
        a = fn(); torch.cuda.synchronize()
        t0 = time.perf_counter(); a = fn(); torch.cuda.synchronize(); dt = time.perf_counter() - t0
        if cert: keep = a
        else: print("   identical:", bool(torch.equal(keep, a)))
        print("cost %d %dx%d certify %d: %.2f ms, path %s" % (cost, k, k, cert, dt * 1e3, names[ctx.last_path()]), flush=True)
