"""One kernel family under rocprofv3 (argument: sad | ssd | ncc | generic13).  GPU box only."""
import sys, numpy as np, torch
sys.path.insert(0, ".")
from visionworkbench_amd import stereo, synth
from visionworkbench_amd.core import BBox2i
what = sys.argv[1] if len(sys.argv) > 1 else "sad"
W = 4096
L, R, _ = synth.stereo_pair(W, W, 129, 1)
Lg, Rg = torch.from_numpy(L).cuda(), torch.from_numpy(R).cuda()
cost, k = {"sad": (0, 7), "ssd": (1, 7), "ncc": (2, 11), "generic13": (0, 13)}[what]
for _ in range(3):
    stereo.calc_disparity(cost, Lg, Rg, BBox2i(0, 0, W, W), (129, 1), (k, k))
torch.cuda.synchronize()
