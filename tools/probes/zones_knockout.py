#!/usr/bin/env python3
"""Knock-out timing of the zone matcher's big launches (GPU box; `make -C visionworkbench_amd/csrc stamps`): parts of the kernel are switched
off (bits: 1 patch reads of the horizontal pass, 2 its plane writes, 4 the plane reads of the vertical pass, 8 the compare chain,
16 the barrier, 32 the right-precision loads) and the level-0 launches timed.  Results are wrong by construction; only level 0 is touched,
so the work of the launches stays the same.  usage: python tools/zones_knockout.py [prefilter cost kernel]"""
import ctypes, os, sys, collections
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("VWGPU_LIBRARY", os.path.join(ROOT, "tools", "build", "libvwgpu_stamps.so"))
import torch  # noqa: E402
from visionworkbench_amd import _lib, core, stereo, synth  # noqa: E402
from visionworkbench_amd.core import BBox2i  # noqa: E402

pf, cost, k = [int(v) for v in sys.argv[1:4]] if len(sys.argv) > 3 else (2, 2, 11)
lib = _lib.load()
lib.vwgpu_debug_set_zone_knock.argtypes = [ctypes.c_int]
L, R, _ = synth.stereo_pair(4096, 4096, 129, 1)
Lg, Rg = torch.from_numpy(L).cuda(), torch.from_numpy(R[:, 64:64 + 4096].copy()).cuda()
ctx = core.default_context(0)
run = lambda: stereo.pyramid_correlate(Lg, Rg, None, None, pf, 1.4 if pf else 0.0, BBox2i.from_corners((-64, -1), (64, 1)), (k, k), cost,
                                       consistency_threshold=2, filter_half_kernel=5, max_pyramid_levels=5, bbox=BBox2i(256, 256, 1024, 1024))
run(); run(); torch.cuda.synchronize()
for bits in (0, 1, 2, 3, 4, 8, 12, 16, 32, 7, 15, 31, 63, 0):
    assert lib.vwgpu_debug_set_zone_knock(bits) == 0
    run(); torch.cuda.synchronize()
    ctx.profile_enable(True); ctx.profile_reset()
    run(); torch.cuda.synchronize()
    rec = ctx.profile_read(1 << 16)
    ctx.profile_enable(False)
    z = [ms * 1e3 for n, ms in rec if n == "bm_zones"]
    bx = sum(ms for n, ms in rec if n.startswith("bmx_")) * 1e3
    print("knock %2d: big launches %s us (all bm_zones %.0f us; exact-order kernels %.0f us)" % (bits, " ".join("%.0f" % v for v in z[-2:]), sum(z), bx))
lib.vwgpu_debug_set_zone_knock(0)
