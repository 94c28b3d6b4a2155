"""Randomised check of the "cannot matter" certificate (csrc/bm_zones.hip, ZEdge): pyramid_correlate over pairs whose search range crosses the
image borders, every parameter that enters its premises drawn at random, against the oracle.  GPU box only.  usage: python tools/fuzz_borders.py [cases] [seed]"""
import sys
import numpy as np
sys.path.insert(0, ".")
sys.path.insert(0, "tests")
import visionworkbench_amd as vwa
from visionworkbench_amd import core, stereo
from visionworkbench_amd.core import BBox2i
import oracle

N = int(sys.argv[1]) if len(sys.argv) > 1 else 150
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
ctx = vwa.Context(0)
bad = 0
shares = []
for case in range(N):
    rng = np.random.default_rng(seed * 100000 + case)
    H, W = int(rng.integers(90, 260)), int(rng.integers(120, 340))
    shift = int(rng.integers(-12, 13))
    base = (rng.random((H, W + 96)) * rng.choice([1.0, 180.0, 4000.0]) + rng.choice([0.0, 20.0])).astype(np.float32)
    if rng.random() < 0.7: base = (base + np.roll(base, 1, axis=1) + np.roll(base, 1, axis=0)) / np.float32(3.0)
    if rng.random() < 0.3: base = np.round(base)
    left = base[:, 48:48 + W].copy(); right = base[:, 48 - shift:48 - shift + W].copy()
    cost = int(rng.integers(0, 3)); k = int(rng.choice([3, 5, 7, 9, 11])); kernel = (k, k)
    pf, pfw = [(2, float(np.float32(1.4))), (1, float(np.float32(3.0))), (2, float(np.float32(0.8))), (0, 0.0)][int(rng.integers(0, 4))]
    filt = int(rng.choice([0, 1, 2, 3, 5, 7])); thr = float(rng.choice([-1, 0, 1, 2, 3.5, 12])); levels = int(rng.integers(0, 5))
    reach = int(rng.integers(6, 40)); ry = int(rng.integers(0, 3))
    search = (-reach, -ry, reach + 1, ry + 1)
    lm = rm = None
    if rng.random() < 0.3:
        lm = np.full((H, W), 255, np.uint8); lm[int(H * .1):int(H * .3), :int(W * .12)] = 0
        rm = np.full((H, W), 255, np.uint8); rm[:, -int(W * .08):] = 0
    box = BBox2i.from_corners(search[:2], search[2:])
    ctx.set_option(core.OPT_TRACE, 4)
    try:
        got = stereo.pyramid_correlate(left, right, lm, rm, pf, pfw, box, kernel, cost, 0, 0.0, thr, 0, filt, levels, ctx=ctx)
        shares.append(ctx.get_option(core.OPT_CERT_PERMILLE))
    finally:
        ctx.set_option(core.OPT_TRACE, 0)
    want = oracle.pyramid_correlate(left, right, lm, rm, pf, pfw, search, kernel, cost, 0, 0.0, thr, filt, levels)
    if not np.array_equal(got, want):
        bad += 1
        print("MISMATCH case %d: %d px; H %d W %d shift %d cost %d k %d pf %d filt %d thr %g levels %d reach %d" %
              (case, int((got != want).any(-1).sum()), H, W, shift, cost, k, pf, filt, thr, levels, reach), flush=True)
sh = [s for s in shares if s >= 0]
print("%d cases, %d mismatches; certified share of the %d cases with certified passes: median %d permille, min %d" %
      (N, bad, len(sh), int(np.median(sh)) if sh else -1, min(sh) if sh else -1))
