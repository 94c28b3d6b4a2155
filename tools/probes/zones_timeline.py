#!/usr/bin/env python3
"""Schedule of the zone-matcher launches of one pyramid tile, workgroup by workgroup (GPU box; `make -C visionworkbench_amd/csrc stamps`).
usage: python tools/zones_timeline.py [prefilter cost kernel]   (default 2 2 11 = LoG + NCC 11x11)"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("VWGPU_LIBRARY", os.path.join(ROOT, "tools", "build", "libvwgpu_stamps.so"))
import numpy as np, torch  # noqa: E402
from visionworkbench_amd import _lib, core, stereo, synth  # noqa: E402
from visionworkbench_amd.core import BBox2i  # noqa: E402

pf, cost, k = [int(v) for v in sys.argv[1:4]] if len(sys.argv) > 3 else (2, 2, 11)
lib = _lib.load()
lib.vwgpu_debug_set_zone_stamps.argtypes = [ctypes.c_void_p]
L, R, _ = synth.stereo_pair(4096, 4096, 129, 1)
Lg, Rg = torch.from_numpy(L).cuda(), torch.from_numpy(R[:, 64:64 + 4096].copy()).cuda()
ctx = core.default_context(0)
run = lambda: stereo.pyramid_correlate(Lg, Rg, None, None, pf, 1.4 if pf else 0.0, BBox2i.from_corners((-64, -1), (64, 1)), (k, k), cost,
                                       consistency_threshold=2, filter_half_kernel=5, max_pyramid_levels=5, bbox=BBox2i(256, 256, 1024, 1024))
run(); run(); torch.cuda.synchronize()
buf = torch.zeros((1 << 18) * 4, dtype=torch.int64, device="cuda")
assert lib.vwgpu_debug_set_zone_stamps(ctypes.c_void_p(buf.data_ptr())) == 0
run(); torch.cuda.synchronize()
assert lib.vwgpu_debug_set_zone_stamps(ctypes.c_void_p(0)) == 0
st = buf.cpu().numpy().astype(np.uint64).reshape(-1, 4)
st = st[st[:, 0] != 0]
t0 = st[:, 0].astype(np.float64); t1 = st[:, 1].astype(np.float64)
base = t0.min()
s_us, e_us = (t0 - base) / 100.0, (t1 - base) / 100.0
ev = (st[:, 3] & np.uint64(0xffffffff)).astype(np.float64)
shader_clk = (st[:, 3] >> np.uint64(32)).astype(np.float64)
# launches = clusters of start times: a new launch begins after every workgroup of the previous one has ended
order = np.argsort(s_us)
launches, cur, cur_end = [], [order[0]], e_us[order[0]]
for i in order[1:]:
    if s_us[i] > cur_end + 0.5:
        launches.append(cur); cur = []; cur_end = 0.0
    cur.append(i); cur_end = max(cur_end, e_us[i])
launches.append(cur)
print("prefilter %d cost %d kernel %d: %d workgroups in %d launches" % (pf, cost, k, len(st), len(launches)))
hw = st[:, 2]
cu = ((hw >> np.uint64(32)) & np.uint64(0xf)).astype(np.int64) * 1000 + ((hw >> np.uint64(13)) & np.uint64(7)).astype(np.int64) * 100 + \
     ((hw >> np.uint64(12)) & np.uint64(1)).astype(np.int64) * 50 + ((hw >> np.uint64(8)) & np.uint64(0xf)).astype(np.int64)
for li, idx in enumerate(launches):
    idx = np.array(idx)
    if len(idx) < 200:
        continue
    a, b = s_us[idx].min(), e_us[idx].max()
    life = e_us[idx] - s_us[idx]
    # residency over time
    ts = np.linspace(a, b, 41)
    res = [int(((s_us[idx] <= t) & (e_us[idx] > t)).sum()) for t in ts]
    percu = [int(((cu[idx] == c)).sum()) for c in np.unique(cu[idx])]
    rate = ev[idx] / np.maximum(life, 1e-3)           # evaluations per us of workgroup life
    print("launch %d: %d workgroups, span %.0f us, sum of lives %.0f us (= %.2f resident on average), %.1f M evaluations" %
          (li, len(idx), b - a, life.sum(), life.sum() / (b - a), ev[idx].sum() / 1e6))
    print("   life us p5/50/95/max: %.1f %.1f %.1f %.1f | evaluations per item p5/50/95/max: %.0f %.0f %.0f %.0f | evals per us of life p5/50/95: %.0f %.0f %.0f" %
          (*np.percentile(life, [5, 50, 95]), life.max(), *np.percentile(ev[idx], [5, 50, 95]), ev[idx].max(), *np.percentile(rate, [5, 50, 95])))
    mhz = shader_clk[idx] / np.maximum(life, 1e-3)
    print("   shader clock over a workgroup's life: p5 %.0f  median %.0f  p95 %.0f MHz" % tuple(np.percentile(mhz, [5, 50, 95])))
    print("   resident workgroups over the span (40 samples): " + " ".join(str(r) for r in res))
    print("   workgroups per CU: min %d max %d (%d CUs)" % (min(percu), max(percu), len(percu)))
    pl = ((st[idx, 3] >> np.uint64(52)) & np.uint64(0xfff)).astype(np.float64) / 100.0
    pr = ((st[idx, 3] >> np.uint64(40)) & np.uint64(0xfff)).astype(np.float64) / 100.0
    steps = np.maximum(ev[idx] / 1024.0, 1.0)
    print("   phases (us, p50 / p90): start -> left patch requested %.1f / %.1f | -> first right patch staged %.1f / %.1f | rest of the life (loop) %.1f / %.1f" %
          (np.percentile(pl, 50), np.percentile(pl, 90), np.percentile(pr, 50), np.percentile(pr, 90), np.percentile(life - pr, 50), np.percentile(life - pr, 90)))
    starts = np.sort(s_us[idx]) - a
    print("   start times: #256 %.1f  #512 %.1f  #1024 %.1f  last %.1f us" % tuple(starts[min(k_, len(starts) - 1)] for k_ in (255, 511, 1023, len(starts) - 1)))
