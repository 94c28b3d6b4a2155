"""Randomised cross-check of calc_disparity_sgm (GPU) against the CPU oracle.  The bounded, seeded version runs under
pytest -m gpu (tests/test_fuzz_gpu.py); this is the long-running aid for the GPU box.
usage: python tools/fuzz_sgm_vs_oracle.py [cases] [seed] [mgm]"""
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

N = int(sys.argv[1]) if len(sys.argv) > 1 else 100
SEED = int(sys.argv[2]) if len(sys.argv) > 2 else 1
MGM = len(sys.argv) > 3 and sys.argv[3] == "mgm"
bad = 0
for c in fuzz_cases.sgm_cases(N, SEED):
    h, w = c["left"].shape
    kw = dict(subpixel_mode=c["sub"], search_buffer=(2, 2), memory_limit_mb=c["mem"], left_mask=c["lm"], right_mask=c["rm"],
              prev_disparity=c["prev"], with_subpixel=True, use_mgm=MGM)
    try:
        gi, gs = stereo.calc_disparity_sgm(c["cost"], c["left"], c["right"], BBox2i(0, 0, w, h), c["search"], (c["k"], c["k"]), **kw)
        oi, os_ = oracle.calc_disparity_sgm(c["cost"], c["left"], c["right"], c["search"], c["k"], subpixel=c["sub"], search_buffer=(2, 2),
                                            memory_limit_mb=c["mem"], left_mask=c["lm"], right_mask=c["rm"], prev_disparity=c["prev"], use_mgm=MGM)
    except Exception as e:  # noqa: BLE001
        print("ERROR it=%d: %s" % (c["it"], e)); bad += 1; continue
    ok = np.array_equal(gi, oi) and (gs is None or np.abs(gs - os_).max() < 1e-5)
    if not ok:
        bad += 1
        print("MISMATCH it=%d k=%d cost=%d s=%s img=%dx%d sub=%d mem=%d  n=%d" % (c["it"], c["k"], c["cost"], c["search"], w, h, c["sub"], c["mem"],
                                                                                   int((gi != oi).any(-1).sum())))
print("cases %d%s, mismatches %d" % (N, " (MGM)" if MGM else "", bad))
sys.exit(1 if bad else 0)
