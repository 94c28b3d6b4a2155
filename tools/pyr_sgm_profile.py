import sys, time, collections
import numpy as np, torch
sys.path.insert(0, ".")
from visionworkbench_amd import stereo, synth, core
from visionworkbench_amd.core import BBox2i
W = 4096
ALG = int(sys.argv[1]) if len(sys.argv) > 1 else 1          # 1 = VW_CORRELATION_SGM, 2 = _MGM, 3 = _FINAL_MGM
L, R, _ = synth.stereo_pair(W, W, 129, 1)
Lg, Rg = torch.from_numpy(L).cuda(), torch.from_numpy(R[:, 64:64 + W].copy()).cuda()
ctx = core.default_context(0)
run = lambda: stereo.pyramid_correlate(Lg, Rg, None, None, 0, 0.0, BBox2i.from_corners((-64, -1), (64, 1)), (7, 7), 3, consistency_threshold=2,
                                       filter_half_kernel=5, max_pyramid_levels=5, algorithm=ALG, bbox=BBox2i(1024, 1024, 1024, 1024))
run(); torch.cuda.synchronize()
t0 = time.perf_counter(); run(); torch.cuda.synchronize(); print("algorithm %d: wall %.1f ms" % (ALG, (time.perf_counter() - t0) * 1e3))
ctx.profile_enable(True); ctx.profile_reset(); run(); torch.cuda.synchronize()
agg = collections.OrderedDict()
for n, ms in ctx.profile_read(1 << 14):
    a = agg.setdefault(n, [0, 0.0]); a[0] += 1; a[1] += ms
for n, (c, ms) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:14]: print("%-24s x%3d %.2f ms" % (n, c, ms))
