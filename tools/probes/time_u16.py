"""16-bit imagery at the headline size: packed-u16 SAD kernel vs the float64 kernel.  GPU box only."""
import sys, time
import numpy as np, torch
sys.path.insert(0, ".")
from visionworkbench_amd import stereo, synth, core
from visionworkbench_amd.core import BBox2i
W = 4096
L, R, _ = synth.stereo_pair(W, W, 129, 1)
Lg, Rg = torch.from_numpy(L * 200.0).cuda(), torch.from_numpy(R * 200.0).cuda()      # integers up to 51000
ctx = core.default_context(0)
for path, name in ((core.PATH_NONE, "auto"), (core.PATH_GENERIC_F64, "generic")):
    ctx.force_path(path)
    fn = lambda: stereo.calc_disparity(0, Lg, Rg, BBox2i(0, 0, W, W), (129, 1), (7, 7), ctx=ctx)
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3): fn()
    torch.cuda.synchronize(); ms = (time.perf_counter() - t0) / 3 * 1e3
    ctx.profile_reset(); ctx.profile_enable(True); fn(); torch.cuda.synchronize(); ctx.profile_enable(False)
    print("%s: %.3f ms path=%d  %s" % (name, ms, ctx.last_path(), " ".join("%s=%.3f" % (n, m) for n, m in ctx.profile_read(64))))
ctx.force_path(core.PATH_NONE)
