"""Randomised cross-check of pyramid_correlate with VW_CORRELATION_SGM (GPU) against the CPU oracle: validity identical, values
within 1e-5 (the tolerance north_star grants sub-pixel values).  The bounded, seeded version runs under pytest -m gpu
(tests/test_fuzz_gpu.py).  usage: python tools/fuzz_pyramid_sgm_vs_oracle.py [cases] [seed] [algorithm: 1 SGM, 2 MGM, 3 FINAL_MGM]"""
import sys
import numpy as np
sys.path.insert(0, ".")
sys.path.insert(0, "tests")
import fuzz_cases
import oracle
from visionworkbench_amd import stereo
from visionworkbench_amd.core import BBox2i

import os
if os.environ.get("SGM_PATH_MODE"):      # VWGPU_OPT_SGM_PATH_MODE of the default context (e.g. 16 / 128: four / two lines per wavefront on ragged boxes)
    from visionworkbench_amd import core
    core.default_context(0).set_option(core.OPT_SGM_PATH_MODE, int(os.environ["SGM_PATH_MODE"]))

N = int(sys.argv[1]) if len(sys.argv) > 1 else 60
SEED = int(sys.argv[2]) if len(sys.argv) > 2 else 1
ALG = int(sys.argv[3]) if len(sys.argv) > 3 else 1
bad = 0
for c in fuzz_cases.pyramid_sgm_cases(N, SEED):
    s = c["search"]
    try:
        g = stereo.pyramid_correlate(c["left"], c["right"], c["lm"], c["rm"], 0, 0.0, BBox2i.from_corners(s[:2], s[2:]), (c["k"], c["k"]), c["cost"],
                                     consistency_threshold=c["thr"], min_consistency_level=c["mcl"], filter_half_kernel=c["filt"],
                                     max_pyramid_levels=c["levels"], algorithm=ALG, bbox=None if c["bbox"] is None else BBox2i(*c["bbox"]))
        o = oracle.pyramid_correlate_sgm(c["left"], c["right"], c["lm"], c["rm"], s, c["k"], c["cost"], c["thr"], c["mcl"], c["filt"], c["levels"],
                                         bbox=c["bbox"], algorithm=ALG)
    except Exception as e:  # noqa: BLE001
        print("ERROR it=%d: %s" % (c["it"], e)); bad += 1; continue
    if not (np.array_equal(g[..., 2], o[..., 2]) and np.abs(g[..., :2] - o[..., :2]).max() < 1e-5):
        bad += 1
        print("MISMATCH it=%d img=%s search=%s k=%d cost=%d thr=%g mcl=%d filt=%d levels=%d masks=%s bbox=%s" %
              (c["it"], c["left"].shape, s, c["k"], c["cost"], c["thr"], c["mcl"], c["filt"], c["levels"], c["lm"] is not None, c["bbox"]))
print("cases %d (algorithm %d), mismatches %d" % (N, ALG, bad))
