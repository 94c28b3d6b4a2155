#!/usr/bin/env python3
"""The tile loop of bench.py (4096^2 pair in 16 tiles of 1024^2, pyramid_correlate with 5 levels, +-64 x +-1, L/R check, filters) with the tiles
handed over in GROUPS (vwgpu_pyramid_correlate_batch_dev): sweep of tile threads x group size for SAD 7x7 and LoG 1.4 + NCC 11x11.
Usage: python tools/tile_loop_batch.py [size] [TxG ...]   (size: side of the pair, default 4096; TxG: tile threads x group size)"""
import os
import sys
import threading
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import visionworkbench_amd as vwa  # noqa: E402
from visionworkbench_amd import stereo, synth  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
left, right, _ = synth.stereo_pair(N, N, 129, 1)
lt = torch.from_numpy(left).cuda()
rc = torch.from_numpy(np.ascontiguousarray(right[:, 64:64 + N])).cuda()
tiles = [vwa.BBox2i(x, y, 1024, 1024) for y in range(0, N, 1024) for x in range(0, N, 1024)]
search = vwa.BBox2i.from_corners((-64, -1), (64, 1))
dev = lt.device


def loop(T, G, pf, pw, cost, kk, reps=int(os.environ.get('TLB_REPS', '4'))):
    ctxs = [vwa.Context(dev.index) for _ in range(T)]
    for c in ctxs: c.set_option(18, int(os.environ.get('TLB_TILE16', '2')))      # VWGPU_OPT_ZONE_TILE16
    streams = [torch.cuda.Stream(device=dev) for _ in range(T)]
    groups = [tiles[i:i + G] for i in range(0, len(tiles), G)]
    outs = {}

    def work(t):
        with torch.cuda.stream(streams[t]):
            for gi in range(t, len(groups), T):
                if G == 1:
                    o = [stereo.pyramid_correlate(lt, rc, None, None, pf, pw, search, (kk, kk), cost, consistency_threshold=2, filter_half_kernel=5,
                                                  max_pyramid_levels=5, bbox=groups[gi][0], ctx=ctxs[t])]
                else:
                    o = stereo.pyramid_correlate_batch(lt, rc, None, None, pf, pw, search, (kk, kk), cost, groups[gi], consistency_threshold=2,
                                                       filter_half_kernel=5, max_pyramid_levels=5, ctx=ctxs[t])
                outs[gi] = o
    best = None
    for rep in range(reps + 1):
        torch.cuda.synchronize(dev); t0 = time.perf_counter()
        th = [threading.Thread(target=work, args=(t,)) for t in range(T)]
        [x.start() for x in th]; [x.join() for x in th]
        torch.cuda.synchronize(dev); dt = time.perf_counter() - t0
        if rep: best = dt if best is None else min(best, dt)
    # launches of one pass (profiled kernels only: fills and copies are not counted)
    for c in ctxs: c.profile_reset(); c.profile_enable(True)
    th = [threading.Thread(target=work, args=(t,)) for t in range(T)]
    [x.start() for x in th]; [x.join() for x in th]
    torch.cuda.synchronize(dev)
    nl, ms = 0, 0.0
    agg = {}
    for c in ctxs:
        c.profile_enable(False)
        rec = c.profile_read(1 << 16)
        nl += len(rec); ms += sum(m for _, m in rec)
        for nm, m in rec:
            a = agg.setdefault(nm, [0, 0.0]); a[0] += 1; a[1] += m
    if os.environ.get("TLB_KERNELS"):
        for nm, (cnt, m) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:14]:
            print("      %-26s %5d launches  %8.3f ms  (%.1f us each)" % (nm, cnt, m, m / cnt * 1e3))
    flat = [o for gi in sorted(outs) for o in outs[gi]]
    if os.environ.get("TLB_TRACE"):                      # host-side timeline of one more pass on stderr
        from visionworkbench_amd import core
        for c in ctxs: c.set_option(core.OPT_TRACE, 1)
        th = [threading.Thread(target=work, args=(t,)) for t in range(T)]
        [x.start() for x in th]; [x.join() for x in th]
        torch.cuda.synchronize(dev)
        for c in ctxs: c.set_option(core.OPT_TRACE, 0)
    for c in ctxs: c.close()
    return best, nl / len(tiles), ms / len(tiles), flat


for label, pf, pw, cost, kk in (("SAD 7x7", 0, 0.0, 0, 7), ("LoG 1.4 + NCC 11x11", 2, 1.4, 2, 11)):
    if os.environ.get("TLB_ONLY") and not label.startswith(os.environ["TLB_ONLY"]): continue
    ref = None
    combos = [tuple(int(v) for v in a.split("x")) for a in sys.argv[2:]] or [(4, 1), (4, 4), (2, 8), (1, 16), (4, 2), (2, 4), (8, 2), (3, 6)]
    for T, G in combos:
        if G > len(tiles): continue
        best, lpt, kms, flat = loop(T, G, pf, pw, cost, kk)
        same = ""
        if ref is None: ref = flat
        else: same = " identical to T=4,G=1: %s" % all(torch.equal(a, b) for a, b in zip(ref, flat))
        print("%-22s threads %d x groups of %2d: %7.2f ms per pair  %8.1f Mpix/s  %5.1f profiled launches / tile  kernels %.3f ms / tile%s"
              % (label, T, G, best * 1e3, N * N / best / 1e6, lpt, kms, same), flush=True)
