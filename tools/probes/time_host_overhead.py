"""Host-side cost per calc_disparity call (what bounds small strips): wrapper vs bare C ABI call."""
import sys, time, ctypes
import numpy as np, torch
sys.path.insert(0, ".")
from visionworkbench_amd import stereo, synth, core
from visionworkbench_amd.core import BBox2i
W, rows = 4096, 517
L, R, _ = synth.stereo_pair(W, W, 129, 1)
l, r = torch.from_numpy(L[:rows]).cuda(), torch.from_numpy(R[:rows]).cuda()
ctx = core.default_context(0); lib = ctx._lib
out = torch.empty((rows - 6, W - 6, 3), dtype=torch.int32, device="cuda")
ctx.set_stream(torch.cuda.current_stream().cuda_stream)
def bare():
    return lib.vwgpu_calc_disparity_dev(ctx._h, 0, l.data_ptr(), W, rows, W, r.data_ptr(), r.shape[1], rows, r.shape[1], 7, 7, 129, 1, out.data_ptr(), 0)
def wrapped():
    return stereo.calc_disparity(0, l, r, BBox2i(0, 0, W, rows), (129, 1), (7, 7))
for name, f in (("bare C ABI", bare), ("python wrapper", wrapped)):
    for _ in range(10): f()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(200): f()
    t_issue = (time.perf_counter() - t0) / 200 * 1e6
    torch.cuda.synchronize()
    t_total = (time.perf_counter() - t0) / 200 * 1e6
    print("%-15s issue %.1f us/call, incl. drain %.1f us/call" % (name, t_issue, t_total))
ctx.profile_enable(True); ctx.profile_reset()
for _ in range(20): bare()
torch.cuda.synchronize()
import collections
agg = collections.OrderedDict()
for n, ms in ctx.profile_read(4096):
    a = agg.setdefault(n, [0, 0.0]); a[0] += 1; a[1] += ms
for n, (c, ms) in agg.items(): print("  %-24s %.1f us avg" % (n, ms / c * 1e3))
