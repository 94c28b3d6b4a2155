"""Round 6: the one-direction-per-launch SGM schedule under VWGPU_OPT_SGM_PATH_MODE (lines per workgroup, census costs formed in the
path kernel or read from a volume): every mode against the oracle on a small scene, then timings at 2048 x 2054 x 129.  GPU box only.
usage: python tools/sgm_path_modes.py [modes ...]   (mode = value of the option; default: a sweep)"""
import sys, time
import numpy as np, torch
sys.path.insert(0, ".")
import oracle
from visionworkbench_amd import stereo, synth, core
from visionworkbench_amd.core import BBox2i
OPT = 19
modes = [int(a) for a in sys.argv[1:]] or [32 + 1, 32, 0, 8, 512, 2048]
ctx = core.default_context(0)
# correctness: 2 small scenes x every mode
for (W, H, SX) in ((200, 90, 40), (131, 77, 128)):
    L, R, _ = synth.stereo_pair(W, H, SX + 1, 1, block=32)
    oi, _ = oracle.calc_disparity_sgm(3, L, R, (SX, 0), 7)
    for m in modes:
        ctx.set_option(OPT, m)
        gi = stereo.calc_disparity_sgm(3, L, R, BBox2i(0, 0, W, H), (SX, 0), (7, 7), ctx=ctx)
        ok = np.array_equal(gi, oi)
        print("scene %dx%d D=%d mode %2d: %s" % (W, H, SX + 1, m, "identical" if ok else "DIFFERENT (%d px)" % int((gi != oi).any(axis=-1).sum())), flush=True)
W, H, SX = 2048, 2054, 128
L, R, _ = synth.stereo_pair(W, H, SX + 1, 1)
Lg, Rg = torch.from_numpy(L).cuda(), torch.from_numpy(R).cuda()
ref = None
for m in modes:
    ctx.set_option(OPT, m)
    run = lambda: stereo.calc_disparity_sgm(3, Lg, Rg, BBox2i(0, 0, W, H), (SX, 0), (7, 7), with_subpixel=True, memory_limit_mb=200000, ctx=ctx)
    run(); torch.cuda.synchronize()
    ts = []
    for _ in range(3):
        t0 = time.perf_counter(); out = run(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    ctx.profile_enable(True); ctx.profile_reset()
    out = run(); torch.cuda.synchronize()
    rec = dict(ctx.profile_read(1 << 12)); ctx.profile_enable(False)
    o = [x.cpu().numpy() if hasattr(x, "cpu") else np.asarray(x) for x in (out if isinstance(out, tuple) else (out,))]
    if ref is None: ref = o
    same = all(np.array_equal(a, b) for a, b in zip(o, ref))
    print("mode %2d: wall %.2f ms, paths %.2f ms, cost %.2f ms, wta %.2f ms  %s" % (m, min(ts) * 1e3, rec.get("sgm_paths", 0), rec.get("sgm_cost", 0),
          rec.get("sgm_wta", 0), "same as first mode" if same else "DIFFERS from first mode"), flush=True)
