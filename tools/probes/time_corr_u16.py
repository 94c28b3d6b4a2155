"""SSD / NCC on 12-bit integer imagery at 4096^2 x 129 disparities: packed kernel (bm_corr_u16) vs the float64 kernel, per-kernel
times from HIP events.  GPU box only.  usage: python tools/time_corr_u16.py [width]"""
import sys, time
import numpy as np, torch
sys.path.insert(0, ".")
from visionworkbench_amd import stereo, synth, core
from visionworkbench_amd.core import BBox2i
W = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
L, R, _ = synth.stereo_pair(W, W, 129, 1)
L = np.floor(L * 16.0 + 7.0).astype(np.float32); R = np.floor(R * 16.0 + 7.0).astype(np.float32)     # [7, 4087]
Lg, Rg = torch.from_numpy(L).cuda(), torch.from_numpy(R).cuda()
box = BBox2i(0, 0, W, W)
ctx = core.default_context(0)
res = {}
for path in (core.PATH_DOT_U16, core.PATH_GENERIC_F64):
    ctx.force_path(path)
    for cost, k in [(1, 7), (2, 7), (1, 9), (2, 9), (1, 11), (2, 11), (2, 5)]:
        fn = lambda: stereo.calc_disparity(cost, Lg, Rg, box, (129, 1), (k, k), ctx=ctx)
        a = fn(); torch.cuda.synchronize()
        n = 5 if path == core.PATH_DOT_U16 else 1
        t0 = time.perf_counter()
        for _ in range(n): fn()
        torch.cuda.synchronize(); ms = (time.perf_counter() - t0) / n * 1e3
        ctx.profile_reset(); ctx.profile_enable(True); fn(); torch.cuda.synchronize(); ctx.profile_enable(False)
        rec = ctx.profile_read(64)
        same = ""
        if path == core.PATH_DOT_U16: res[(cost, k)] = a
        else: same = " identical=%s" % bool(torch.equal(a, res[(cost, k)]))
        print("path=%d cost=%d k=%d: %.3f ms last_path=%d  %s%s" % (path, cost, k, ms, ctx.last_path(), " ".join("%s=%.3f" % (n_, m) for n_, m in rec), same), flush=True)
ctx.force_path(core.PATH_NONE)
