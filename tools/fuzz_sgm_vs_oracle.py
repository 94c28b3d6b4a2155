"""Randomised cross-check of calc_disparity_sgm (GPU) against the CPU oracle: random sizes, 1-D / 2-D searches, kernels, cost
types, masks, previous-level disparities, memory limits, sub-pixel modes.  Development aid for the GPU box (the oracle is
test infrastructure; nothing here ships)."""
import sys
import numpy as np
sys.path.insert(0, ".")
import oracle
from visionworkbench_amd import stereo
from visionworkbench_amd.core import BBox2i

N = int(sys.argv[1]) if len(sys.argv) > 1 else 100
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
bad = 0
for it in range(N):
    k = int(rng.choice([3, 5, 7, 9]))
    cost = int(rng.choice([3, 4]))
    sx = int(rng.integers(0, 70)); sy = int(rng.choice([0, 0, 1, 2, 6]))
    if rng.random() < 0.2:
        sx = int(rng.integers(100, 140)); sy = 0
    h = int(rng.integers(k + 2, 70)); w = int(rng.integers(k + 2, 120))
    base = rng.random((h + sy + 8, w + sx + 8))
    scale = float(rng.choice([255.0, 1.0, 4000.0]))
    base = (base * scale).astype(np.float32)
    d0 = (int(rng.integers(0, sx + 1)), int(rng.integers(0, sy + 1)))
    left = np.ascontiguousarray(base[4:4 + h, 4:4 + w])
    right = np.ascontiguousarray(base[4 - min(d0[1], 4):4 - min(d0[1], 4) + h + sy, 4 - min(d0[0], 4):4 - min(d0[0], 4) + w + sx])
    oh, ow = h - k + 1, w - k + 1
    lm = rm = prev = None
    if rng.random() < 0.4:
        lm = np.full((oh, ow), 255, np.uint8)
        y0, x0 = int(rng.integers(0, oh)), int(rng.integers(0, ow))
        lm[y0:y0 + 9, x0:x0 + 14] = 0
    if rng.random() < 0.3:
        rm = np.full((oh + sy, ow + sx), 255, np.uint8)
        rm[:, -int(rng.integers(1, 6)):] = 0
    if rng.random() < 0.4:
        prev = np.zeros(((oh + 1) // 2, (ow + 1) // 2, 3), np.int32)
        prev[..., 0] = rng.integers(0, sx // 2 + 1, prev.shape[:2])
        prev[..., 1] = rng.integers(0, sy // 2 + 1, prev.shape[:2])
        prev[..., 2] = np.where(rng.random(prev.shape[:2]) < 0.85, np.iinfo(np.int32).max, 0)
    sub = int(rng.choice([0, 1, 2, 3, 4, 5])) if (sx > 0 or sy > 0) else 0
    mem = int(rng.choice([6000, 6000, 1]))
    kw = dict(subpixel_mode=sub, search_buffer=(2, 2), memory_limit_mb=mem, left_mask=lm, right_mask=rm, prev_disparity=prev, with_subpixel=True)
    try:
        gi, gs = stereo.calc_disparity_sgm(cost, left, right, BBox2i(0, 0, w, h), (sx, sy), (k, k), **kw)
        oi, os_ = oracle.calc_disparity_sgm(cost, left, right, (sx, sy), k, subpixel=sub, search_buffer=(2, 2), memory_limit_mb=mem,
                                            left_mask=lm, right_mask=rm, prev_disparity=prev)
    except Exception as e:  # noqa: BLE001
        print("ERROR it=%d: %s" % (it, e)); bad += 1; continue
    ok = np.array_equal(gi, oi) and (gs is None or np.abs(gs - os_).max() < 1e-5)
    if not ok:
        bad += 1
        print("MISMATCH it=%d k=%d cost=%d s=%dx%d img=%dx%d masks=%s/%s prev=%s sub=%d mem=%d  n=%d" %
              (it, k, cost, sx, sy, w, h, lm is not None, rm is not None, prev is not None, sub, mem, int((gi != oi).any(-1).sum())))
print("cases %d, mismatches %d" % (N, bad))
sys.exit(1 if bad else 0)
