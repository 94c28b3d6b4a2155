"""Per-kernel launch counts / device time inside one pyramid_correlate tile (vwgpu_profile_*), for the input classes that
pick different matchers: integer imagery (tile kernels), LoG / mean-subtracted imagery and NCC (exact-order kernels when a
level's box sums could round).  GPU box only.  usage: python tools/pyr_profile.py [tile] [prefilter cost kernel]..."""
import sys, time, collections
import numpy as np, torch
sys.path.insert(0, ".")
from visionworkbench_amd import stereo, synth, core
from visionworkbench_amd.core import BBox2i
W = 4096
tile = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
cases = [(0, 0, 7), (2, 0, 7), (1, 0, 7), (0, 2, 11), (2, 2, 11), (0, 1, 7)]
L, R, _ = synth.stereo_pair(W, W, 129, 1)
Lg, Rg = torch.from_numpy(L).cuda(), torch.from_numpy(R[:, 64:64 + W].copy()).cuda()
ctx = core.default_context(0)
import os
ctx.set_option(core.OPT_EXACT_SPLIT, int(os.environ.get("PYR_EXACT_SPLIT", "0")))      # tool-side switch
ONLY = os.environ.get("PYR_ONLY", "")
ctx.set_option(core.OPT_ZONE_SXC, int(os.environ.get("PYR_ZONE_SXC", "0")))
ctx.set_option(core.OPT_CERT_F32, int(os.environ.get("PYR_CERT_F32", "1")))                 # 0: the certified pass in float64 only (round 4)
TRACE = int(os.environ.get("PYR_TRACE", "0"))                     # 2: the shape of every level's work on stderr, 4: certification counts
for pf, cost, k in cases:
    if ONLY and ONLY != "%d,%d,%d" % (pf, cost, k): continue
    args = dict(consistency_threshold=2, filter_half_kernel=5, max_pyramid_levels=5, bbox=BBox2i(256, 256, tile, tile))
    run = lambda: stereo.pyramid_correlate(Lg, Rg, None, None, pf, 1.4 if pf else 0.0, BBox2i.from_corners((-64, -1), (64, 1)), (k, k), cost, **args)
    run(); torch.cuda.synchronize()
    if os.environ.get("PYR_TRACE_PLAIN"): ctx.set_option(core.OPT_TRACE, int(os.environ["PYR_TRACE_PLAIN"]))    # the host timeline without the profiler's events
    t0 = time.perf_counter(); out = run(); torch.cuda.synchronize(); wall = time.perf_counter() - t0
    ctx.set_option(core.OPT_TRACE, 0)
    ctx.profile_enable(True); ctx.profile_reset()
    ctx.set_option(core.OPT_TRACE, TRACE)
    run(); torch.cuda.synchronize()
    ctx.set_option(core.OPT_TRACE, 0)
    rec = ctx.profile_read(1 << 16)
    ctx.profile_enable(False)
    agg = collections.OrderedDict()
    for n, ms in rec:
        a = agg.setdefault(n, [0, 0.0]); a[0] += 1; a[1] += ms
    print("== prefilter %d cost %d kernel %dx%d tile %d^2: wall %.2f ms; kernels %.2f ms; valid %.3f" %
          (pf, cost, k, k, tile, wall * 1e3, sum(v[1] for v in agg.values()), float((out[..., 2] != 0).float().mean())))
    for n, (c, ms) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:12]:
        print("   %-28s launches %6d  device %.3f ms" % (n, c, ms))
    if os.environ.get("PYR_LAUNCHES"):              # every launch of the matchers in stream order (coarsest level first)
        print("   launches (us): " + " ".join("%s=%.0f" % (n.replace("bm_zones", "z").replace("bmx_", "x"), ms * 1e3) for n, ms in rec
                                               if os.environ["PYR_LAUNCHES"] == "all" or n.startswith("bm_zones") or n.startswith("bmx_") or n == "zone_precision"))
