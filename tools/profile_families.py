"""One pass over every kernel family for `rocprofv3 --kernel-trace --stats` (profiles/r01e_families.md).  GPU box only."""
import sys
import numpy as np, torch
sys.path.insert(0, ".")
from visionworkbench_amd import stereo, synth
from visionworkbench_amd.core import BBox2i
W = 4096
L, R, _ = synth.stereo_pair(W, W, 129, 1)
Lg, Rg = torch.from_numpy(L).cuda(), torch.from_numpy(R).cuda()
Rc = Rg[:, 64:64 + W].contiguous()
box = BBox2i(0, 0, W, W)
for rep in range(3):
    stereo.calc_disparity(0, Lg, Rg, box, (129, 1), (7, 7))                    # config 2: packed SAD
    stereo.calc_disparity(1, Lg, Rg, box, (129, 1), (7, 7))                    # SSD, dot path
    d = stereo.calc_disparity(2, Lg, Rg, box, (129, 1), (11, 11))              # config 3: NCC 11x11, dot path
    df = torch.zeros((W, W, 3), dtype=torch.float32, device="cuda")
    df[5:5 + d.shape[0], 5:5 + d.shape[1]] = d.float()
    df[..., 2] = (df[..., 2] != 0).float()
    stereo.parabola_subpixel(df, Lg, Rg, 0, 0.0, (11, 11))                     # config 3: parabola
    search = BBox2i.from_corners((-64, -1), (64, 1))
    stereo.pyramid_correlate(Lg, Rc, None, None, 2, 1.4, search, (7, 7), 0, consistency_threshold=2, filter_half_kernel=5,
                             max_pyramid_levels=5, bbox=BBox2i(1024, 1024, 1024, 1024))          # BM pyramid tile, LoG prefilter
    stereo.pyramid_correlate(Lg, Rc, None, None, 2, 1.4, search, (11, 11), 2, consistency_threshold=2, filter_half_kernel=5,
                             max_pyramid_levels=5, bbox=BBox2i(1024, 1024, 1024, 1024))          # the correlate tool's defaults: LoG + NCC 11x11 (exact-order kernels)
    stereo.pyramid_correlate(Lg, Rc, None, None, 0, 0.0, search, (7, 7), 0, consistency_threshold=2, filter_half_kernel=5,
                             max_pyramid_levels=5, bbox=BBox2i(1024, 1024, 1024, 1024))          # integer imagery, SAD (float32 window sums)
    stereo.pyramid_correlate(Lg, Rc, None, None, 0, 0.0, search, (7, 7), 3, consistency_threshold=2, filter_half_kernel=5,
                             max_pyramid_levels=5, algorithm=1, bbox=BBox2i(1024, 1024, 1024, 1024))   # SGM pyramid tile
    stereo.calc_disparity_sgm(3, Lg[:2048, :2048].contiguous(), Rg[:2048, :2048 + 128].contiguous(), BBox2i(0, 0, 2048, 2048), (128, 0), (7, 7),
                              with_subpixel=True, memory_limit_mb=200000)       # config 4 building block: single-level SGM, D = 129
torch.cuda.synchronize()
print("done")
