import sys, numpy as np, torch
sys.path.insert(0, '.')
import visionworkbench_amd as vwa
from visionworkbench_amd import stereo, synth, core
ctx = vwa.Context(0)
def run(w, h, sx, seeds=(10, 11, 12), reps=3):
    left, right, _ = synth.stereo_pair(w, h, sx, 1, seeds=seeds)
    lt, rt = torch.from_numpy(left).cuda(), torch.from_numpy(right).cuda()
    ctx.profile_reset(); ctx.profile_enable(True)
    for _ in range(reps):
        stereo.calc_disparity(0, lt, rt, vwa.bounding_box(left), (sx, 1), (7, 7), ctx=ctx)
    torch.cuda.synchronize(); ctx.profile_enable(False)
    recs = ctx.profile_read()
    d = {}
    for k, v in recs: d.setdefault(k, []).append(v * 1e3)
    print(w, h, sx, {k: round(min(v), 1) for k, v in d.items()})
for (w, h) in [(1030, 22), (2054, 22), (4096, 22), (4096, 38), (1030, 262), (4096, 262), (4096, 4096)]:
    run(w, h, 129)
