"""PCIe-inclusive rate of the host-pointer entry point (vwgpu_calc_disparity): H2D + kernels + D2H, config 2."""
import sys, time, numpy as np
sys.path.insert(0, '.')
import visionworkbench_amd as vwa
from visionworkbench_amd import stereo, synth
left, right, _ = synth.stereo_pair(4096, 4096, 129, 1)
ctx = vwa.Context(0)
for _ in range(2):
    out = stereo.calc_disparity(0, left, right, vwa.bounding_box(left), (129, 1), (7, 7), ctx=ctx)
t0 = time.perf_counter(); n = 5
for _ in range(n):
    out = stereo.calc_disparity(0, left, right, vwa.bounding_box(left), (129, 1), (7, 7), ctx=ctx)
dt = (time.perf_counter() - t0) / n
print("host-pointer entry (pageable numpy buffers): %.2f ms/call = %.0f Mpix/s" % (dt * 1e3, 4090 * 4090 / dt / 1e6))
