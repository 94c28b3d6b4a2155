"""Fused SGM sweeps vs one direction per launch (OPT_SGM_SWEEP) on the GPU box: identical disparities, timings."""
import sys, time
import numpy as np, torch
sys.path.insert(0, ".")
from visionworkbench_amd import stereo, synth, core
from visionworkbench_amd.core import BBox2i
ctx = core.default_context(0)
def run(W, H, SX, sweep, k=7, prof=False):
    L, R, _ = synth.stereo_pair(W, H, SX + 1, 1)
    Lg, Rg = torch.from_numpy(L).cuda(), torch.from_numpy(R).cuda()
    ctx.set_option(core.OPT_SGM_SWEEP, 1 if sweep else 0)
    f = lambda: stereo.calc_disparity_sgm(3, Lg, Rg, BBox2i(0, 0, W, H), (SX, 0), (k, k), with_subpixel=True, memory_limit_mb=200000, ctx=ctx)
    out = f(); torch.cuda.synchronize()
    t0 = time.perf_counter(); out = f(); torch.cuda.synchronize(); wall = time.perf_counter() - t0
    rec = []
    if prof:
        ctx.profile_enable(True); ctx.profile_reset(); f(); torch.cuda.synchronize(); rec = ctx.profile_read(4096); ctx.profile_enable(False)
    return out[0].cpu().numpy(), out[1].cpu().numpy(), wall, rec
for (W, H, SX) in [(64, 20, 16), (300, 37, 128), (257, 130, 128), (1000, 300, 60), (2048, 2048, 128)]:
    a = run(W, H, SX, False)
    b = run(W, H, SX, True, prof=True)
    same = np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
    print("%dx%d D=%d: identical=%s  per-direction %.2f ms, sweeps %.2f ms  %s" % (W, H, SX + 1, same, a[2] * 1e3, b[2] * 1e3,
          " ".join("%s=%.2f" % (n, m) for n, m in b[3] if n in ("sgm_paths", "sgm_wta", "sgm_cost"))), flush=True)
    if not same:
        d = (a[0] != b[0]).any(-1)
        print("   mismatching pixels:", int(d.sum()), "first rows/cols:", np.argwhere(d)[:5].tolist())
