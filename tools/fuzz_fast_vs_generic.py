"""Randomised cross-check of the packed-u8 matchers (SAD; with cost 1 / 2 as third argument: the SSD / NCC dot kernel) against the generic float64 kernel (itself pinned to the oracle by
tests/test_bm_gpu.py): random sizes, kernels, 1-D / 2-D searches, flat patches, both matcher flavours.  GPU box only."""
import os, sys
import numpy as np, torch
sys.path.insert(0, ".")
import visionworkbench_amd as vwa
from visionworkbench_amd import core, stereo

N = int(sys.argv[1]) if len(sys.argv) > 1 else 300
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
COST = int(sys.argv[3]) if len(sys.argv) > 3 else 0
ctx = vwa.Context(0)
kernels = [(3, 3), (5, 5), (7, 7), (7, 5), (9, 9), (11, 11)]
bad = 0
for it in range(N):
    kx, ky = kernels[rng.integers(len(kernels))]
    if COST:
        kx, ky = int(rng.integers(1, 9)) * 2 - 1, int(rng.integers(1, 9)) * 2 - 1
    sx = int(rng.integers(1, 140)); sy = int(rng.choice([1, 1, 1, 2, 3])) if COST == 0 else 1
    w = int(rng.integers(kx, 2600)); h = int(rng.integers(ky, 200))
    lo = 1 if COST == 2 else 0                       # NCC: no all-zero windows (1/0 sends the tile to the generic kernel anyway)
    left = np.floor(rng.random((h, w)) * (256 - lo)).astype(np.float32) + lo
    right = np.floor(rng.random((h + sy - 1, w + sx - 1)) * (256 - lo)).astype(np.float32) + lo
    # paste shifted copies so that there is structure, and flat patches so that validity matters
    d = int(rng.integers(0, sx))
    right[:h, d:d + w] = np.where(rng.random((h, w)) < 0.7, left, right[:h, d:d + w])
    if rng.random() < 0.5:
        y0, x0 = int(rng.integers(0, h)), int(rng.integers(0, w))
        left[y0:y0 + 20, x0:x0 + 40] = 7.0
        right[y0:y0 + 24, x0:x0 + 60 + sx] = 7.0
    os.environ["VWGPU_SAD_SPLIT"] = str(int(rng.integers(0, 2)))
    lt, rt = torch.from_numpy(left).cuda(), torch.from_numpy(right).cuda()
    ctx.force_path(core.PATH_NONE)
    a = stereo.calc_disparity(COST, lt, rt, vwa.bounding_box(left), (sx, sy), (kx, ky), ctx=ctx).cpu().numpy()
    pa = ctx.last_path()
    ctx.force_path(core.PATH_GENERIC_F64)
    b = stereo.calc_disparity(COST, lt, rt, vwa.bounding_box(left), (sx, sy), (kx, ky), ctx=ctx).cpu().numpy()
    ctx.force_path(core.PATH_NONE)
    if not np.array_equal(a, b):
        bad += 1
        print("MISMATCH it=%d k=%dx%d s=%dx%d img=%dx%d path=%d split=%s  n=%d" % (it, kx, ky, sx, sy, w, h, pa, os.environ["VWGPU_SAD_SPLIT"], int((a != b).any(-1).sum())))
print("cases %d, mismatches %d" % (N, bad))
sys.exit(1 if bad else 0)
