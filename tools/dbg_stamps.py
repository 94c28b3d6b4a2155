import sys, numpy as np, torch
sys.path.insert(0, '.')
import visionworkbench_amd as vwa
from visionworkbench_amd import stereo, synth
ctx = vwa.Context(0)
w, h = int(sys.argv[1]), int(sys.argv[2])
left, right, _ = synth.stereo_pair(w, h, 129, 1)
lt, rt = torch.from_numpy(left).cuda(), torch.from_numpy(right).cuda()
for _ in range(2):
    stereo.calc_disparity(0, lt, rt, vwa.bounding_box(left), (129, 1), (7, 7), ctx=ctx)
    torch.cuda.synchronize(); ctx.synchronize()
