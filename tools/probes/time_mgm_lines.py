"""use_mgm on 2-D search boxes (the ragged kernels): the eight passes as lines in one launch (VWGPU_OPT_MGM_SWEEP 0) against one launch per
front (1).  usage: python tools/time_mgm_lines.py [size ...]   GPU box only."""
import sys
import numpy as np, torch
sys.path.insert(0, ".")
from visionworkbench_amd import stereo, synth, core
from visionworkbench_amd.core import BBox2i
sizes = [int(a) for a in sys.argv[1:]] or [256, 512, 1024]
ctx = core.default_context(0)
for W in sizes:
    for SX, SY in ((42, 2), (128, 2)):
        L, R, _ = synth.stereo_pair(W, W, SX + 1, SY + 1)
        Lg, Rg = torch.from_numpy(L).cuda(), torch.from_numpy(R).cuda()
        outs = {}
        for opt in (0, 1):
            ctx.set_option(core.OPT_MGM_SWEEP, opt)
            run = lambda: stereo.calc_disparity_sgm(3, Lg, Rg, BBox2i(0, 0, W, W), (SX, SY), (7, 7), use_mgm=True, with_subpixel=False, memory_limit_mb=200000)
            run(); torch.cuda.synchronize()
            ctx.profile_enable(True); ctx.profile_reset()
            o = run(); torch.cuda.synchronize()
            rec = dict(ctx.profile_read(1 << 12)); ctx.profile_enable(False)
            outs[opt] = (o[0] if isinstance(o, tuple) else o).cpu().numpy()
            print("%4d^2, %3d x %d disparities, %s: sgm_mgm_paths %.2f ms (%.2f us per step of W + H)" %
                  (W, SX + 1, SY + 1, "lines " if opt == 0 else "fronts", rec.get("sgm_mgm_paths", 0.0), rec.get("sgm_mgm_paths", 0.0) * 1e3 / (2 * W)), flush=True)
        ctx.set_option(core.OPT_MGM_SWEEP, 0)
        print("      identical:", np.array_equal(outs[0], outs[1]))
