#!/usr/bin/env python3
"""Replay one case of tests/fuzz_cases.bm_float_cases against the oracle with every schedule: fp32 tier on / off, certificate off.
usage: python tools/replay_float_case.py <n> <seed> <it>   (seed as the generator saw it: campaign seed + tier)"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import fuzz_cases, oracle
import visionworkbench_amd as vwa
from visionworkbench_amd import core, stereo
n, seed, it = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
ctx = vwa.Context(0)
names = {v: k for k, v in vars(core).items() if k.startswith("PATH_")}
for c in fuzz_cases.bm_float_cases(n, seed):
    if c["it"] != it: continue
    L, R = c["left"], c["right"]
    print("case", it, "cost", c["cost"], "kernel", c["kernel"], "search", c["search"], "left", L.shape, "kind", c["kind"],
          "range", float(L.min()), float(L.max()), float(R.min()), float(R.max()))
    want = oracle.calc_disparity(c["cost"], L, R, c["kernel"], c["search"])
    for f32, cert in ((1, 1), (0, 1), (1, 0)):
        ctx.set_option(core.OPT_CERT_F32, f32); ctx.set_option(core.OPT_CERTIFY, cert)
        ctx.set_option(core.OPT_TRACE, 4)
        got = stereo.calc_disparity(c["cost"], L, R, vwa.bounding_box(L), c["search"], c["kernel"], ctx=ctx)
        bad = (got != want).any(-1)
        print("  fp32 tier %d certify %d: path %s, %d mismatching pixels, certified %d per mille, float64 tier %d per mille" %
              (f32, cert, names.get(ctx.last_path()), int(bad.sum()), ctx.get_option(core.OPT_CERT_PERMILLE), ctx.get_option(core.OPT_CERT_F64_PERMILLE)))
        ctx.set_option(core.OPT_TRACE, 0)
        for y, x in list(zip(*np.nonzero(bad)))[:6]:
            print("    (x %d, y %d): got %s want %s" % (x, y, got[y, x].tolist(), want[y, x].tolist()))
    np.savez(os.path.join(ROOT, "gpurun_out", "float_case_%d_%d.npz" % (seed, it)), left=L, right=R, cost=c["cost"], kernel=c["kernel"], search=c["search"])
    break
