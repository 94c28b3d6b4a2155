"""Randomised cross-check of the packed-u8 matchers (SAD; with cost 1 / 2 as third argument: the SSD / NCC dot kernel)
against the generic float64 kernel (itself pinned to the oracle by tests/test_bm_gpu.py).  The bounded, seeded version
of this runs under pytest -m gpu (tests/test_fuzz_gpu.py); this is the long-running aid.  GPU box only.
usage: python tools/fuzz_fast_vs_generic.py [cases] [seed] [cost] [scale]   (scale 4096 / 65536: the packed-u16 kernels)"""
import os, sys
import numpy as np, torch
sys.path.insert(0, ".")
sys.path.insert(0, "tests")
import fuzz_cases
import visionworkbench_amd as vwa
from visionworkbench_amd import core, stereo

N = int(sys.argv[1]) if len(sys.argv) > 1 else 300
SEED = int(sys.argv[2]) if len(sys.argv) > 2 else 1
COST = int(sys.argv[3]) if len(sys.argv) > 3 else 0
SCALE = int(sys.argv[4]) if len(sys.argv) > 4 else 256
paths = {}
ctx = vwa.Context(0)
bad = 0
for c in fuzz_cases.bm_cases(N, SEED, COST, SCALE):
    ctx.set_option(core.OPT_SAD_GROUPS, 1 + int(c["split"]))
    lt, rt = torch.from_numpy(c["left"]).cuda(), torch.from_numpy(c["right"]).cuda()
    ctx.force_path(core.PATH_NONE)
    a = stereo.calc_disparity(COST, lt, rt, vwa.bounding_box(c["left"]), c["search"], c["kernel"], ctx=ctx).cpu().numpy()
    pa = ctx.last_path()
    paths[pa] = paths.get(pa, 0) + 1
    ctx.force_path(core.PATH_GENERIC_F64)
    b = stereo.calc_disparity(COST, lt, rt, vwa.bounding_box(c["left"]), c["search"], c["kernel"], ctx=ctx).cpu().numpy()
    ctx.force_path(core.PATH_NONE)
    if not np.array_equal(a, b):
        bad += 1
        print("MISMATCH it=%d k=%s s=%s img=%s path=%d split=%s  n=%d" % (c["it"], c["kernel"], c["search"], c["left"].shape, pa, c["split"], int((a != b).any(-1).sum())))
print("cases %d, mismatches %d, paths %s" % (N, bad, sorted(paths.items())))
sys.exit(1 if bad else 0)
