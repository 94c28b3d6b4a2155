"""Tile throughput of pyramid_correlate with several host threads per GPU — how the reference runs it (block_write_image pulls
tiles with a thread pool, src/vw/Image/ImageIO.h:228-251; one engine context = one stream per thread).  A tile's level loop
syncs with the host once per level (the zone tree of the next level is built on the host); with a few tiles in flight those
gaps are filled by the other tiles' kernels.  GPU box only.  usage: python tools/pyr_throughput.py [threads...]"""
import sys, time, threading
import numpy as np, torch
sys.path.insert(0, ".")
from visionworkbench_amd import stereo, synth, core
from visionworkbench_amd.core import BBox2i
W, tile = 4096, 1024
L, R, _ = synth.stereo_pair(W, W, 129, 1)
Lg, Rg = torch.from_numpy(L).cuda(), torch.from_numpy(R[:, 64:64 + W].copy()).cuda()
tiles = [(x, y) for y in range(0, W, tile) for x in range(0, W, tile)]
cases = [("SAD 7x7, integer imagery", 0, 0, 7), ("NCC 11x11, integer imagery", 0, 2, 11), ("LoG 1.4 + SAD 7x7", 2, 0, 7),
         ("LoG 1.4 + NCC 11x11", 2, 2, 11)]
threads = [int(a) for a in sys.argv[1:]] or [1, 2, 4, 8]
import os
SPLIT = int(os.environ.get("PYR_EXACT_SPLIT", "0"))      # tool-side switch -> core.OPT_EXACT_SPLIT of every context
SCRATCH = int(os.environ.get("PYR_EXACT_SCRATCH_MB", "4096"))   # -> core.OPT_EXACT_SCRATCH_MB (zone groups small enough to stay in the last-level cache?)
ONLY = os.environ.get("PYR_ONLY", "")
torch.cuda.synchronize()
for name, pf, cost, k in cases:
    # device time of one tile's kernels, for reference
    ctx0 = core.default_context(0)
    if ONLY and ONLY not in name: continue
    ctx0.set_option(core.OPT_EXACT_SPLIT, SPLIT)
    ctx0.set_option(core.OPT_EXACT_SCRATCH_MB, SCRATCH)
    run0 = lambda c, x, y: stereo.pyramid_correlate(Lg, Rg, None, None, pf, 1.4 if pf else 0.0, BBox2i.from_corners((-64, -1), (64, 1)), (k, k), cost,
                                                    consistency_threshold=2, filter_half_kernel=5, max_pyramid_levels=5, bbox=BBox2i(x, y, tile, tile), ctx=c)
    run0(ctx0, 1024, 1024); torch.cuda.synchronize()
    ctx0.profile_enable(True); ctx0.profile_reset(); run0(ctx0, 1024, 1024); torch.cuda.synchronize()
    kern = sum(ms for _, ms in ctx0.profile_read(1 << 16)); ctx0.profile_enable(False)
    line = "%-28s kernels %.2f ms/tile |" % (name, kern)
    for T in threads:
        ctxs = [core.Context(0) for _ in range(T)]
        for c in ctxs: c.set_option(core.OPT_EXACT_SPLIT, SPLIT)
        for c in ctxs: c.set_option(core.OPT_EXACT_SCRATCH_MB, SCRATCH)
        for c in ctxs: run0(c, 0, 0)                        # arenas warm
        torch.cuda.synchronize()
        todo = list(tiles) * 6
        lock = threading.Lock()
        streams = [torch.cuda.Stream() for _ in range(T)]     # the engine launches on the caller's current torch stream
        def work(c, st):
            with torch.cuda.stream(st):
                while True:
                    with lock:
                        if not todo: return
                        x, y = todo.pop()
                    run0(c, x, y)
        c0 = os.times()
        t0 = time.perf_counter()
        th = [threading.Thread(target=work, args=(c, st)) for c, st in zip(ctxs, streams)]
        for t in th: t.start()
        for t in th: t.join()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        c1 = os.times()
        line += " %d thr %.2f ms/tile (cpu %.2f user + %.2f sys ms/tile)" % (T, dt / (6 * len(tiles)) * 1e3, (c1.user - c0.user) / (6 * len(tiles)) * 1e3, (c1.system - c0.system) / (6 * len(tiles)) * 1e3)
        for c in ctxs: c.close()
    print(line, flush=True)
