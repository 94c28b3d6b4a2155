cd $GRAFT_REPO_ROOT
python - <<'PY' 2>&1 | grep -v amdgpu | tail -60
import sys, time
sys.path.insert(0, ".")
import torch
from visionworkbench_amd import stereo, synth, core
from visionworkbench_amd.core import BBox2i
L, R, _ = synth.stereo_pair(4096, 4096, 129, 1)
Lg, Rg = torch.from_numpy(L).cuda(), torch.from_numpy(R[:, 64:64 + 4096].copy()).cuda()
ctx = core.default_context(0)
run = lambda: stereo.pyramid_correlate(Lg, Rg, None, None, 0, 0.0, BBox2i.from_corners((-64, -1), (64, 1)), (7, 7), 0,
                         consistency_threshold=2, filter_half_kernel=5, max_pyramid_levels=5, bbox=BBox2i(256, 256, 1024, 1024))
run(); run(); torch.cuda.synchronize()
ctx.set_option(core.OPT_TRACE, 1)
t0 = time.perf_counter(); run(); torch.cuda.synchronize(); print("wall %.2f ms" % ((time.perf_counter() - t0) * 1e3))
ctx.set_option(core.OPT_TRACE, 0)
import cProfile, pstats
t0 = time.perf_counter()
for _ in range(20): run()
torch.cuda.synchronize(); print("20 tiles: %.2f ms each" % ((time.perf_counter() - t0) * 50))
import os
c0 = os.times(); t0 = time.perf_counter()
for _ in range(50): run()
torch.cuda.synchronize(); c1 = os.times()
print("50 tiles one thread: wall %.2f ms/tile, cpu user %.2f sys %.2f ms/tile" % ((time.perf_counter() - t0) * 20, (c1.user - c0.user) * 20, (c1.system - c0.system) * 20))
PY
