"""Randomised cross-check of pyramid_correlate (GPU, host entry) against the CPU oracle.  The bounded, seeded version runs
under pytest -m gpu (tests/test_fuzz_gpu.py); this is the long-running aid for the GPU box.
usage: python tools/fuzz_pyramid_vs_oracle.py [cases] [seed] [float_scene_probability | corner] [prefilters, e.g. 0,1,2]"""
import sys
import numpy as np
sys.path.insert(0, ".")
sys.path.insert(0, "tests")
import fuzz_cases
import oracle
from visionworkbench_amd import stereo
from visionworkbench_amd.core import BBox2i

N = int(sys.argv[1]) if len(sys.argv) > 1 else 60
SEED = int(sys.argv[2]) if len(sys.argv) > 2 else 1
CORNER = len(sys.argv) > 3 and sys.argv[3] == "corner"
FLOATP = float(sys.argv[3]) if len(sys.argv) > 3 and not CORNER else 0.0
PF = tuple(int(v) for v in sys.argv[4].split(",")) if len(sys.argv) > 4 else (0,)
bad = 0
for c in (fuzz_cases.pyramid_corner_cases(N, SEED) if CORNER else fuzz_cases.pyramid_cases(N, SEED, prefilters=PF, float_scene=FLOATP)):
    s = c["search"]
    oracle.set_blob_filter_area(c.get("blob", 0))
    try:
        g = stereo.pyramid_correlate(c["left"], c["right"], c["lm"], c["rm"], c["pf"], c["pfw"], BBox2i.from_corners(s[:2], s[2:]), c["kernel"],
                                     c["cost"], 0, 0.0, c["thr"], 0, c["filt"], c["levels"], bbox=None if c["bbox"] is None else BBox2i(*c["bbox"]), blob_filter_area=c.get("blob", 0))
        o = oracle.pyramid_correlate(c["left"], c["right"], c["lm"], c["rm"], c["pf"], c["pfw"], s, c["kernel"], c["cost"], 0, 0.0, c["thr"],
                                     c["filt"], c["levels"], bbox=c["bbox"])
    except Exception as e:  # noqa: BLE001
        print("ERROR it=%d: %s" % (c["it"], e)); bad += 1; continue
    if not np.array_equal(g, o):
        bad += 1
        print("MISMATCH it=%d img=%s search=%s k=%s cost=%d thr=%g filt=%d levels=%d masks=%s bbox=%s pf=%d float=%s n=%d" %
              (c["it"], c["left"].shape, s, c["kernel"], c["cost"], c["thr"], c["filt"], c["levels"], c["lm"] is not None, c["bbox"], c["pf"], c["is_float"],
               int((g != o).any(-1).sum())))
print("cases %d, mismatches %d" % (N, bad))
sys.exit(1 if bad else 0)
