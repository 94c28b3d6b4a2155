"""Strong-scaling proxy on one GPU: time of calc_disparity on 1/N row strips of config 2 (what each of N GPUs runs)."""
import sys, time
import numpy as np, torch
sys.path.insert(0, ".")
from visionworkbench_amd import stereo, synth, core
from visionworkbench_amd.core import BBox2i
W = 4096
L, R, _ = synth.stereo_pair(W, W, 129, 1)
Lg, Rg = torch.from_numpy(L).cuda(), torch.from_numpy(R).cuda()
base = None
for n in (1, 2, 4, 8):
    rows = (W - 6) // n + 6
    l, r = Lg[:rows].contiguous(), Rg[:rows].contiguous()
    f = lambda: stereo.calc_disparity(0, l, r, BBox2i(0, 0, W, rows), (129, 1), (7, 7))
    for _ in range(5): f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(100): f()
    torch.cuda.synchronize(); ms = (time.perf_counter() - t0) / 100 * 1e3
    base = base or ms
    print("N=%d rows=%d: %.3f ms/step  -> speed-up %.2fx" % (n, rows, ms, base / ms))
