#!/usr/bin/env python3
"""Single-level calc_disparity on the float-texture twin of the bench pair (4096^2, search 129x1): wall ms per call and kernel times for the
default dispatch, VWGPU_OPT_CERTIFY = 0 (exact-order kernels) and the forced bm_generic kernel.  Usage: python tools/time_float_single.py [size]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import visionworkbench_amd as vwa  # noqa: E402
from visionworkbench_amd import core, stereo, synth  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
left, right, _ = synth.stereo_pair(N, N, 129, 1)
rng = np.random.default_rng(20260926)
lf = torch.from_numpy((left * np.float32(0.37) + rng.random(left.shape, dtype=np.float32)).astype(np.float32)).cuda()
rf = torch.from_numpy((right * np.float32(0.37) + rng.random(right.shape, dtype=np.float32)).astype(np.float32)).cuda()
ctx = vwa.Context(0)
bb = vwa.BBox2i(0, 0, N, N)
names = {v: k for k, v in vars(core).items() if k.startswith("PATH_")}


def run(label, cost, k, reps=3):
    fn = lambda: stereo.calc_disparity(cost, lf, rf, bb, (129, 1), (k, k), ctx=ctx)
    out = fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        out = fn()
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / reps * 1e3
    ctx.profile_reset(); ctx.profile_enable(True)
    fn(); torch.cuda.synchronize()
    ctx.profile_enable(False)
    per = {}
    for name, ms in ctx.profile_read(1 << 12):
        per[name] = per.get(name, 0.0) + ms
    px = (N - k + 1) ** 2
    print("%-34s cost %d %2dx%-2d  %8.3f ms  %8.1f Mpix/s  path %-18s %s" % (label, cost, k, k, wall, px / wall / 1e3, names.get(ctx.last_path()),
                                                                           {a: round(b, 3) for a, b in per.items()}), flush=True)
    return out


for cost, k in ((0, 7), (1, 7), (2, 11)):
    a = run("default", cost, k)
    ctx.set_option(core.OPT_CERT_F32, 0)
    a64 = run("fp32 tier off", cost, k)
    ctx.set_option(core.OPT_CERT_F32, 1)
    assert torch.equal(a, a64)
    ctx.set_option(core.OPT_CERTIFY, 0)
    b = run("certify off", cost, k, reps=1)
    ctx.set_option(core.OPT_CERTIFY, 1)
    ctx.force_path(core.PATH_GENERIC_F64)
    c = run("forced bm_generic (not exact here)", cost, k, reps=1)
    ctx.force_path(core.PATH_NONE)
    print("   default == certify off: %s;  pixels where bm_generic differs: %d" % (bool(torch.equal(a, b)), int((a != c).any(-1).sum())), flush=True)
