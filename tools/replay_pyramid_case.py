"""Replays case <index> of tools/fuzz_pyramid_vs_oracle.py <seed> and lists the pixels where the GPU tile and the oracle differ.
GPU box only."""
import sys
import numpy as np
sys.path.insert(0, ".")
import oracle
from visionworkbench_amd import stereo
from visionworkbench_amd.core import BBox2i
target = int(sys.argv[1]); seed = int(sys.argv[2])
rng = np.random.default_rng(seed)
for it in range(target + 1):
    H, W = int(rng.integers(90, 360)), int(rng.integers(120, 520))
    left = np.floor(rng.random((H, W)) * 256).astype(np.float32)
    right = np.empty_like(left)
    band = int(rng.integers(20, 90))
    for y0 in range(0, H, band):
        right[y0:y0 + band] = np.roll(left[y0:y0 + band], int(rng.integers(-7, 8)), axis=1)
    if rng.random() < 0.5:
        right = np.roll(right, int(rng.integers(-2, 3)), axis=0)
    mx, my = int(rng.integers(1, 12)), int(rng.integers(0, 4))
    search = (-mx, -my, mx + int(rng.integers(0, 3)), my + 1)
    k = int(rng.choice([3, 5, 7, 9])); ky = int(rng.choice([k, k, 5]))
    cost = int(rng.choice([0, 0, 1]))
    thr = float(rng.choice([-1, 1, 2]))
    filt = int(rng.choice([0, 3, 5])); levels = int(rng.integers(0, 5))
    lm = rm = None
    if rng.random() < 0.4:
        lm = np.full(left.shape, 255, np.uint8); rm = np.full(right.shape, 255, np.uint8)
        y0, x0 = int(rng.integers(0, H)), int(rng.integers(0, W))
        lm[y0:y0 + 30, x0:x0 + 50] = 0
        rm[:, -int(rng.integers(1, 40)):] = 0
    bbox = None
    if rng.random() < 0.6:
        bw, bh = int(rng.integers(24, min(200, W))), int(rng.integers(24, min(160, H)))
        bbox = (int(rng.integers(0, W - bw + 1)), int(rng.integers(0, H - bh + 1)), bw, bh)
print("case", target, W, H, search, k, ky, cost, thr, filt, levels, lm is not None, bbox)
for t in (thr, -1.0):
    g = stereo.pyramid_correlate(left, right, lm, rm, 0, 0.0, BBox2i.from_corners(search[:2], search[2:]), (k, ky), cost, 0, 0.0, t, 0, filt, levels)
    o = oracle.pyramid_correlate(left, right, lm, rm, 0, 0.0, search, (k, ky), cost, 0, 0.0, t, filt, levels)
    d = np.argwhere((g != o).any(-1))
    print("thr", t, "mismatching pixels", len(d))
    for (y, x) in d[:12]:
        print("  (x=%d,y=%d) gpu %s oracle %s  lmask %s" % (x, y, g[y, x], o[y, x], lm[y, x] if lm is not None else None))
if lm is not None:
    ys, xs = np.where(lm == 0); print("lmask zero block rows %d..%d cols %d..%d" % (ys.min(), ys.max(), xs.min(), xs.max()))
    print("rmask zero cols from", np.where(rm[0] == 0)[0].min())

