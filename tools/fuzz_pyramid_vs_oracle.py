"""Randomised cross-check of pyramid_correlate (GPU, host entry: window staging, zone scheduler on the device table) against
the CPU oracle: random scenes, search boxes, kernels, SAD / SSD, L/R thresholds, filter radii, level counts, masks, prefilter
NONE, interior / border tiles.  Development aid for the GPU box."""
import sys
import numpy as np
sys.path.insert(0, ".")
import oracle
from visionworkbench_amd import stereo
from visionworkbench_amd.core import BBox2i

N = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
bad = 0
for it in range(N):
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
    try:
        g = stereo.pyramid_correlate(left, right, lm, rm, 0, 0.0, BBox2i.from_corners(search[:2], search[2:]), (k, ky), cost, 0, 0.0, thr, 0,
                                     filt, levels, bbox=None if bbox is None else BBox2i(*bbox))
        o = oracle.pyramid_correlate(left, right, lm, rm, 0, 0.0, search, (k, ky), cost, 0, 0.0, thr, filt, levels, bbox=bbox)
    except Exception as e:  # noqa: BLE001
        print("ERROR it=%d: %s" % (it, e)); bad += 1; continue
    if not np.array_equal(g, o):
        bad += 1
        print("MISMATCH it=%d img=%dx%d search=%s k=%dx%d cost=%d thr=%g filt=%d levels=%d masks=%s bbox=%s n=%d" %
              (it, W, H, search, k, ky, cost, thr, filt, levels, lm is not None, bbox, int((g != o).any(-1).sum())))
print("cases %d, mismatches %d" % (N, bad))
sys.exit(1 if bad else 0)
