"""bm_sad_u8 kernel time vs number of workgroups (latency curve).  GPU box only."""
import sys, time
import numpy as np, torch
sys.path.insert(0, ".")
from visionworkbench_amd import stereo, synth, core
from visionworkbench_amd.core import BBox2i
W = 4096
L, R, _ = synth.stereo_pair(W, W, 129, 1)
Lg, Rg = torch.from_numpy(L).cuda(), torch.from_numpy(R).cuda()
ctx = core.default_context(0)
for width in (262, 1030, 4096):
  for tiles in (1, 2, 4, 8, 16, 32, 33, 48, 64, 128, 256):
    rows = tiles * 16 + 6
    l, r = Lg[:rows, :width].contiguous(), Rg[:rows, :width + 128].contiguous()
    f = lambda: stereo.calc_disparity(0, l, r, BBox2i(0, 0, width, rows), (129, 1), (7, 7))
    for _ in range(3): f()
    ctx.profile_enable(True); ctx.profile_reset()
    for _ in range(10): f()
    torch.cuda.synchronize()
    t = [ms for n, ms in ctx.profile_read(4096) if n == "bm_sad_u8"]
    ctx.profile_enable(False)
    wgs = tiles * ((width - 6 + 255) // 256)
    print("width %4d row-tiles %3d -> %4d WGs: %.1f us  (%.2f us per WG-slot round of 512)" % (width, tiles, wgs, np.mean(t) * 1e3, np.mean(t) * 1e3 / max(1, -(-wgs // 512))))
