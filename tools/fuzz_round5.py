#!/usr/bin/env python3
"""Randomised campaign for the round-5 paths against the oracle: single-level calc_disparity on float rasters (certified pass, fp32 tier,
exact-order fallback) and pyramid_correlate_batch (tile groups).  usage: python tools/fuzz_round5.py [n_single] [n_batch] [seed]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import fuzz_cases  # noqa: E402
import oracle  # noqa: E402
import visionworkbench_amd as vwa  # noqa: E402
from visionworkbench_amd import core, stereo  # noqa: E402
from visionworkbench_amd.core import BBox2i  # noqa: E402

n1 = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
n2 = int(sys.argv[2]) if len(sys.argv) > 2 else 300
seed = int(sys.argv[3]) if len(sys.argv) > 3 else 5000
ctx = vwa.Context(0)
names = {v: k for k, v in vars(core).items() if k.startswith("PATH_")}
t0 = time.time()
bad, paths = [], {}
for tier in (1, 0):
    ctx.set_option(core.OPT_CERT_F32, tier)
    for c in fuzz_cases.bm_float_cases(n1 if tier else n1 // 4, seed + tier):
        got = stereo.calc_disparity(c["cost"], c["left"], c["right"], vwa.bounding_box(c["left"]), c["search"], c["kernel"], ctx=ctx)
        p = names.get(ctx.last_path())
        paths[(tier, p)] = paths.get((tier, p), 0) + 1
        want = oracle.calc_disparity(c["cost"], c["left"], c["right"], c["kernel"], c["search"])
        if not np.array_equal(got, want):
            bad.append(("single", tier, c["it"], int((got != want).any(-1).sum())))
    for c in fuzz_cases.bm_float_corner_cases((n1 if tier else n1 // 4) // 2, seed + tier):
        got = stereo.calc_disparity(c["cost"], c["left"], c["right"], vwa.bounding_box(c["left"]), c["search"], c["kernel"], ctx=ctx)
        p = names.get(ctx.last_path())
        paths[(tier, p)] = paths.get((tier, p), 0) + 1
        want = oracle.calc_disparity(c["cost"], c["left"], c["right"], c["kernel"], c["search"])
        if not np.array_equal(got, want):
            bad.append(("corner", tier, c["it"], c["cost"], c["kernel"], c["search"], c["kind"], int((got != want).any(-1).sum())))
ctx.set_option(core.OPT_CERT_F32, 1)
print("single-level float rasters: %d + %d cases (fp32 tier on / off) + half as many corner cases each, %d mismatches, paths %s, %.0f s" % (n1, n1 // 4, len(bad), paths, time.time() - t0), flush=True)
t0 = time.time()
tiles = 0
for c in fuzz_cases.batch_cases(n2, seed + 7):
    s = c["search"]
    got = stereo.pyramid_correlate_batch(c["left"], c["right"], c["lm"], c["rm"], c["pf"], c["pfw"], BBox2i.from_corners(s[:2], s[2:]), c["kernel"], c["cost"],
                                         [BBox2i(*b) for b in c["boxes"]], consistency_threshold=c["thr"], filter_half_kernel=c["filt"],
                                         max_pyramid_levels=c["levels"], ctx=ctx)
    for b, g in zip(c["boxes"], got):
        tiles += 1
        o = oracle.pyramid_correlate(c["left"], c["right"], c["lm"], c["rm"], c["pf"], c["pfw"], s, c["kernel"], c["cost"], 0, 0.0, c["thr"], c["filt"], c["levels"], bbox=b)
        if not np.array_equal(g, o):
            bad.append(("batch", c["it"], b, int((g != o).any(-1).sum())))
print("pyramid_correlate_batch: %d scenes, %d tiles, %d mismatches so far, %.0f s" % (n2, tiles, len(bad), time.time() - t0), flush=True)
print("MISMATCHES: %s" % bad if bad else "no mismatch")
sys.exit(1 if bad else 0)
